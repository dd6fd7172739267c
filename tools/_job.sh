cd $GRAFT_REPO_ROOT
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --only-leg config3_sampler 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps(d.get('config3_sampler'))[:900])"
