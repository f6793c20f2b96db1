#!/bin/bash
# tools/variant_sweep.sh <workload> <spp> : frame time of every build variant under _diag/lib_*.so (compile-time tuning A/B),
# after the in-tree library.  Run on the GPU box.
W=${1:-mesh_1m}; S=${2:-256}
echo "default: $(python tools/option_sweep.py --workload $W --spp $S 2>/dev/null | grep total)"
for L in _diag/lib_*.so; do
  echo "$(basename $L): $(TRHIP_LIB=$PWD/$L python tools/option_sweep.py --workload $W --spp $S 2>/dev/null | grep total)"
done
