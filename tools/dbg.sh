#!/bin/bash
# temporary: our command line on the multi fixture with -r5 -Z, output kept for a diff against the golden
O=gpurun_out/dbg; mkdir -p $O /tmp/dbgm
for f in genome.sfx reads.fa; do gunzip -c tests/golden/multi/$f.gz > /tmp/dbgm/$f; done
biokanga_amd/bin/biokanga align -i /tmp/dbgm/reads.fa -I /tmp/dbgm/genome.sfx -o $O/r5R5ZmB.m6.sam -M6 -s3 -r5 -R5 -Z mB > $O/r5R5ZmB.log 2>&1
biokanga_amd/bin/biokanga align -i /tmp/dbgm/reads.fa -I /tmp/dbgm/genome.sfx -o $O/r5R3XzmA.m5.sam -M5 -s3 -r5 -R3 -X -z '^ma$' > $O/r5R3XzmA.log 2>&1
biokanga_amd/bin/biokanga align -i /tmp/dbgm/reads.fa -I /tmp/dbgm/genome.sfx -o $O/r5R5k20x3Z.m4.bed -M4 -s3 -r5 -R5 -k20 -x3 -Z mB > $O/r5R5k20x3Z.log 2>&1
gzip -f $O/*.sam $O/*.bed; ls -la $O
