cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 && timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 && bash scripts/gpu_c1.sh && timeout 300 python bench.py --steps 20 --warmup 3 --cpu-budget 6 2>/dev/null | cut -c1-1500
