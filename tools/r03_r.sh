cd $GRAFT_REPO_ROOT
for w in 1 0 3; do echo "wide=$w"; python tools/attn_bench.py --check --iters 50 --wide $w 2>/dev/null | grep -v HuggingFace; done
