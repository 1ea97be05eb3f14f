#!/bin/bash
# local helper: build both libraries, run the pb gate on the GPU box, print the summary
cd /root/repo
make -C glia_amd/csrc -j8 2>&1 | grep -E " error|warning: var"
make -C glia_amd/csrc prof 2>&1 | grep -E " error"
/usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/_gpu4.sh' > gpurun_out/_gate.log 2>&1
tail -9 gpurun_out/_gate.log | cut -c1-260
grep -h sha1 gpurun_out/r02d/pb512_hash.txt gpurun_out/r02d/pb256_hash.txt | sort -u
grep "wide phases" gpurun_out/r02d/pb1024_prof.txt | tail -1
