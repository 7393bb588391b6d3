mkdir -p gpurun_out
python - <<'PY' > gpurun_out/r04e_load_segment_variants_bits.txt 2>&1
import sys; sys.path.insert(0, ".")
import torch, gemm_hls_amd as g
dev = torch.device("cuda:0")
for dtype, tdt, knob in (("half", torch.float16, "f16_variant"), ("uint8_t", torch.uint8, "i8_variant")):
    for (n, k, m) in ((1000, 4096, 1032), (4096, 8192, 4096)):
        a = torch.empty((n, k), dtype=tdt, device=dev); b = torch.empty((k, m), dtype=tdt, device=dev)
        g._check(g.lib().mm_fill_device(0, g.DTYPES[dtype], a.data_ptr(), a.numel(), 3)); g._check(g.lib().mm_fill_device(0, g.DTYPES[dtype], b.data_ptr(), b.numel(), 4))
        if dtype == "half": a.mul_(2.0 ** -6)
        g.set_tuning(knob, 200); ref = g.matmul(a, b, dtype).clone()
        for v in (201, 202, 203, 204, 205):
            g.set_tuning(knob, v)
            name = g.kernel_name(g.make_config(dtype), n, k, m)
            same = all(bool(torch.equal(g.matmul(a, b, dtype), ref)) for _ in range(5))
            print(dtype, (n, k, m), v, name, "bit-identical to 200:", same, flush=True)
        g.set_tuning(knob, -1)
PY
cat gpurun_out/r04e_load_segment_variants_bits.txt | tail -22
timeout 600 python tools/sweep.py f16 --variants 200,201,202,203,204,205,200,201,202,203,204,205 --sizes 16384,32768 --reps 5 > gpurun_out/r04e_f16_load_segment_variants.txt 2>&1; cat gpurun_out/r04e_f16_load_segment_variants.txt
timeout 600 python tools/sweep.py uint8 --variants 200,201,202,203,204,205,200,201,202,203,204,205 --sizes 16384,32768 --reps 5 > gpurun_out/r04e_i8_load_segment_variants.txt 2>&1; cat gpurun_out/r04e_i8_load_segment_variants.txt
