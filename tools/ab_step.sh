# same-box A/B of library variants on the headline step and the batch sweep: bash tools/ab_step.sh NAME1 NAME2 ...  ("main" = the product build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" == "main" ]; then unset SAMPLENET_AMD_LIB; else export SAMPLENET_AMD_LIB=$PWD/tools/_ab/libsamplenet_hip_$v.so; fi
    python bench.py --steps 1500 --warmup 100 --no-probes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$v', 'B=32', round(d['value']), round(d['ms_per_step']*1e3,2), 'us')"
  done
done
for v in "$@"; do
  if [ "$v" == "main" ]; then unset SAMPLENET_AMD_LIB; else export SAMPLENET_AMD_LIB=$PWD/tools/_ab/libsamplenet_hip_$v.so; fi
  python tools/batch_sweep.py ${SWEEP:-512 2048} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$v', 'B=%d'%d['batch'], round(d['clouds_per_s']), round(d['ms_per_step'],4),'ms')"
done
