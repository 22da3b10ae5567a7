#!/usr/bin/env bash
# Developer aid: one bench line per (label, environment, arguments) — step time and the kernel groups' ms per step.
#   bash tools/sweep_knobs.sh          (on the GPU box)
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  out=$(env "${envs[@]}" python bench.py --no_cpu_baseline --no_vgg_variant --no_gen_metric --steps 12 --warmup 3 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d.get('kernels',{})
def g(n): return k.get(n,{}).get('ms_per_step',0)
print(d['ms_per_step'], 'norm', round(g('norm_bwd_reduce')+g('norm_bwd_dx')+g('norm_stats')+g('norm_apply_fwd'),2), 'igemm', round(g('igemm_fwd')+g('igemm_fwd64')+g('igemm_wgrad'),2), 'wino', round(g('wino_conv')+g('wino4_conv')+g('wino_wgrad'),2))")
  echo "$label: $out"
}
if [ "$#" -gt 0 ]; then "$@"; exit 0; fi
run C3 X=1 --
run C4 X=1 -- --config C4 --batch 4
run C2 X=1 -- --config C2 --image_size 128
run C5 X=1 -- --config C5 --batch 6
run C4-old CSG_WINO_MIN_PIXELS=16384 -- --config C4 --batch 4
run C5-old CSG_WINO_MIN_PIXELS=16384 -- --config C5 --batch 6
