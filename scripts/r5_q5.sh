#!/bin/bash
export DS2_EXPERIMENTAL=1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
RING_VARIANTS="0 q q5" python scripts/r5_ring.py check | tail -4
RING_VARIANTS="0 q q5" python scripts/r5_ring.py time
