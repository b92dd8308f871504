#!/bin/bash
# round 5: the A/B builds of the time-serial scan backward (tools/tm_ab.py bwd <names>); leaves the DEFAULT library in place at the end
set -e
cd "$(dirname "$0")/.."
bash tools/build_variant.sh l0 -DAUM_SCANT_LSUM=0
bash tools/build_variant.sh l1 -DAUM_SCANT_LSUM=1
bash tools/build_variant.sh l1a -DAUM_SCANT_LSUM=1 -DAUM_LSUM_PUT_ASM=1
bash tools/build_variant.sh l2 -DAUM_SCANT_LSUM=2
bash tools/build_variant.sh l2a -DAUM_SCANT_LSUM=2 -DAUM_LSUM_PUT_ASM=1
bash tools/build_variant.sh babl1 -DAUM_SCANT_LSUM=0 -DAUM_SCANT_BABL=1
bash tools/build_variant.sh babl2 -DAUM_SCANT_LSUM=0 -DAUM_SCANT_BABL=2
python audio-mamba-aum_amd/csrc/build.py --force > /tmp/build_default.log 2>&1
echo default rebuilt
