mkdir -p gpurun_out
timeout 600 python tools/f32_default_ab.py --rounds 4 > gpurun_out/r04c_f32_default_ab_steady_state.txt 2>&1; cat gpurun_out/r04c_f32_default_ab_steady_state.txt
timeout 300 python tools/kxn_prepass_check.py --sizes 4096,6144,8192,12288 --min-m 0 > gpurun_out/r04c_kxn_prepass_forced_small_m.txt 2>&1; cat gpurun_out/r04c_kxn_prepass_forced_small_m.txt
