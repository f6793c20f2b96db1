#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
for v in libtracehip lib_a lib_b; do
echo == $v
TRHIP_LIB=$PWD/trace.jl_amd/$v.so timeout 300 python tools/film_probe.py --spp 256 2>/dev/null < /dev/null | head -6
done
