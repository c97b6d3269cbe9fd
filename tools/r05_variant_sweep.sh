#!/bin/bash
# round 5, last experiment: compile-time policies A/B'd on the tower alone (tools/lnx_wait_sweep.py: one 8704-image pass per call),
# one process per library build, the builds interleaved over three rounds on one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${OUT:-r05_variants}; mkdir -p $O
V=lossyless_amd/variants
for round in 1 2 3; do
  for lib in ${LIBS:-product split4 split16 gm2 gm8 rmwplain dmaplain}; do
    if [ $lib = product ]; then unset LLA_LIB; else export LLA_LIB=$PWD/$V/liblossyless_amd_$lib.so; fi
    echo "== round $round $lib" >> $O/sweep.txt
    timeout 120 python tools/lnx_wait_sweep.py --waits 6000,24000 --rounds 2 --passes 8 >> $O/sweep.txt 2>> $O/err.txt
  done
done
grep -v "^library" $O/sweep.txt
