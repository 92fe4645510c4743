#!/bin/bash
# Print VGPR/AGPR/SGPR/spill/LDS/occupancy per kernel of the HIP sources (compile-only, no GPU needed).
cd "$(dirname "$0")/../amphibian_vae_latent_detector_amd/csrc"
for f in k_frontend k_encoder k_encoder_split k_encoder_patch k_encoder_bf16; do
  EXTRA=""; case $f in k_encoder_split|k_encoder_patch) EXTRA="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $EXTRA -c $f.hip -o /tmp/$f.res.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur={}
for line in sys.stdin:
    m=re.search(r"remark:\s+(.*?):\s+(\S+) \[-Rpass", line)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=="Function Name": cur={}
    cur[k]=v
    if k.startswith("LDS Size"):
        print("%-60s VGPR %-4s AGPR %-3s SGPR %-3s spill %-3s scratch %-5s occ %-2s LDS %s"%(cur["Function Name"][:60],cur.get("VGPRs"),cur.get("AGPRs"),cur.get("TotalSGPRs"),cur.get("VGPRs Spill"),cur.get("ScratchSize [bytes/lane]"),cur.get("Occupancy [waves/SIMD]"),v))
'
done
