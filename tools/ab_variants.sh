#!/bin/bash
# Several builds of the library on ONE box: ab/libamph_<name>.so copied over the in-tree library in turn, bench.py run with the
# given arguments, value + per-kernel durations printed.   usage: tools/ab_variants.sh "<bench.py arguments>" rounds name [name ...]
L=amphibian_vae_latent_detector_amd/libamph.so
ARGS=$1; ROUNDS=$2; shift 2
cp $L /tmp/libamph_keep.so
for i in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    cp ab/libamph_$v.so $L
    python bench.py $ARGS > /tmp/b.json 2>> gpurun_out/ab.err || { cp /tmp/libamph_keep.so $L; exit 1; }
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print(sys.argv[1], round(d["value"]), [(x["kernel"], round(x["ms_per_launch"] * 1e3, 1)) for x in d["kernels"]], flush=True)
PY
  done
done
cp /tmp/libamph_keep.so $L
