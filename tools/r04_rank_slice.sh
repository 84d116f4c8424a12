#!/bin/bash
# profiles/r04_rank_slice.txt: one rank's share of a G-rank factorisation alone on one MI355X (tools/rank_slice.py), all on one box
R=gpurun_out/r04_rank_slice.txt; : > $R
run() { echo "# $*" >> $R; env "$@" timeout -k 10 200 python tools/rank_slice.py $N $G $g $X 2>&1 | grep -v amdgpu.ids >> $R || exit 1; }
N=65536; X=""
G=1; g=0; run TGP_DIST_FINISH=0
G=1; g=0; run DEFAULTS=1
G=2; g=1; run DEFAULTS=1
G=4; g=3; run DEFAULTS=1
G=8; for g in 0 3 7; do run DEFAULTS=1; done
g=7
run TGP_DIST_FUSED=0
run TGP_DIST_HALF_TILES=0
run TGP_DIST_FINISH=0
run TGP_DIST_FINISH=8
run TGP_DIST_FINISH=32
run TGP_DIST_QUEUE=-1
run TGP_DIST_GROUP=2
X=launches; run DEFAULTS=1; X=""
N=131072; G=8; g=7; run DEFAULTS=1
N=32768; G=8; g=7; run DEFAULTS=1
N=16384; G=4; g=3; run DEFAULTS=1
cat $R | grep "N=\|^#"
