mkdir -p gpurun_out
( for sz in 32768 16384 8192; do timeout 300 python tools/variant_rr.py f16 --variants 200,203,204,205,201 --size $sz --rounds 14; done
  for sz in 32768 16384 8192; do timeout 300 python tools/variant_rr.py uint8 --variants 200,203,204,205,201 --size $sz --rounds 14; done ) > gpurun_out/r04f_load_segment_variants_round_robin.txt 2>&1
cat gpurun_out/r04f_load_segment_variants_round_robin.txt
