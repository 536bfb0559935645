#!/bin/bash
# usage: tools/ab_builds.sh <libA> <libB> [rounds]   -- tools/ab_lib_probe.py alternately under two builds of libgfft on one box
a=$1; b=$2; n=${3:-2}
for i in $(seq $n); do
  GFFT_AB_LIB=$a python tools/ab_lib_probe.py 2>&1 | grep "^\["
  GFFT_AB_LIB=$b python tools/ab_lib_probe.py 2>&1 | grep "^\["
done
