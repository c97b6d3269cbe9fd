#!/bin/bash
# which XCDs do the interleaved masks of the seventh call leave? (tools/ubench/xcc_map: XCC id of every workgroup)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_twelfth; mkdir -p $O
make -C tools/ubench xcc_map > /dev/null 2>&1
m() { python -c "print(','.join(str(i) for i in range(256) if $1))"; }
( for e in "i%8<4" "i%8>=4" "i%8==0" "i<8" "i<16" "i%32<16" "(i//8)%2==0"; do echo "mask bits {i: $e}"; HSA_CU_MASK=0:$(m "$e") tools/ubench/xcc_map | head -1; done ) > $O/xcc_map_interleaved.txt 2>&1
cat $O/xcc_map_interleaved.txt
