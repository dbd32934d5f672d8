#!/bin/bash
# round 5, late: seven waves per SIMD for the rope march (queue of five entries so that seven workgroups fit a CU's LDS) against the shipped six
# build here first: tools/ab_variants.sh build base:"" w7q5:"-DEXA_MARCH_WAVES=7 -DEXA_ROPE_QUEUE=5" w6q5:"-DEXA_ROPE_QUEUE=5"
set -u
cd "${GRAFT_REPO_ROOT:-.}"
tools/ab_variants.sh run
