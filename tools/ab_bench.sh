#!/bin/bash
# A/B on ONE box: ab/libamph_<name>.so copied over the in-tree library in turn, the headline and configs[3] benched.
# usage: tools/ab_bench.sh old new [rounds]
L=amphibian_vae_latent_detector_amd/libamph.so
cp $L /tmp/libamph_keep.so
for i in $(seq 1 ${3:-3}); do
  for v in $1 $2; do
    cp ab/libamph_$v.so $L
    python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-detect --no-extra-legs > /tmp/b.json 2>> gpurun_out/ab.err || exit 1
    python bench.py --preset bench-1s-128 --bf16 --cpu-seconds 0 --no-detect > /tmp/c.json 2>> gpurun_out/ab.err || exit 1
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/b.json")); c = json.load(open("/tmp/c.json"))
k = {x["kernel"]: round(x["ms_per_launch"] * 1e3, 1) for x in d["kernels"]}
kc = {x["kernel"]: round(x["ms_per_launch"] * 1e3, 1) for x in c["kernels"]}
print(sys.argv[1], "headline", round(d["value"]), "cold", round(d["value_cold"]), k.get("stft_mel"), "| config4", round(c["value"]), kc.get("stft_mel"), kc.get("dbnorm"), flush=True)
PY
  done
done
cp /tmp/libamph_keep.so $L
