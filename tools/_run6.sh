cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
python -m pytest tests/test_gpu_attention.py -q -m gpu 2>&1 | tail -5 > gpurun_out/r02f/attn_tests.log
python tools/attn_bench.py --check --iters 50 > gpurun_out/r02f/attn_bench.log 2>&1
SVT_X=1 python - >> gpurun_out/r02f/attn_bench.log 2>&1 <<'PY'
import sys; sys.path.insert(0,'.')
from svt_speechbrain_amd import _lib
_lib.load().svt_debug_set(10, 0)
sys.argv=['attn_bench','--iters','50','--only','base']
exec(open('tools/attn_bench.py').read())
PY
cat gpurun_out/r02f/attn_tests.log gpurun_out/r02f/attn_bench.log
