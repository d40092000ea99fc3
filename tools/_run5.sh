cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "fused_tail or c3_c5 or c4_audio or bf16_mode_error or reload_reaches or b2" 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r02e/tests.log
cat gpurun_out/r02e/tests.log
import json; r=json.load(open('gpurun_out/r02e/$f.json')); print('$f', r['value'], r['ms_per_step'])"; done
