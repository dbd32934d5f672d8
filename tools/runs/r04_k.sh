#!/bin/bash
# round 4: per-kernel times of C5 (4096^2, iso + AO, 16 spp) with inline and with deferred AO rays, and of C3 + iso
O=gpurun_out
for d in 0 1; do
  echo "== C5 ao_defer $d (prepass_split 0: one pre-pass, one march per sample)"
  bash tools/config_timeline.sh $O/r04_k_c5_defer$d --size 4096 --iso 0.5 --ao --spp 16 --steps 2 --warmup 1 --pmc off --option prepass_split=0 --option ao_defer=$d
done
echo "== C3 + iso"
bash tools/config_timeline.sh $O/r04_k_c3iso --config c3_gear --iso 0.5 --steps 10 --pmc off --option prepass_split=0
echo "== C3 + iso + AO, inline / deferred"
for d in 0 1; do bash tools/config_timeline.sh $O/r04_k_c3isoao$d --config c3_gear --iso 0.5 --ao --steps 10 --pmc off --option prepass_split=0 --option ao_defer=$d; done
