#!/bin/bash
# Round-5 decode bundle (run through gpurun): full GPU suite with the flip count stored, kernel stats of the C5 decode, the hop / packed-fp32
# probes, the same-box A/B against the round's first commit, the per-phase timeline, and the default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
SPN_PROFILE_DIR=$O timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -5 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
timeout 120 tools/_bin/hop_probe > $O/hop_probe.txt 2>&1
timeout 60 tools/_bin/pk_probe > $O/pk_probe.txt 2>&1
timeout 900 tools/ab_decode.sh 2 > $O/decode_ab.txt 2>&1
timeout 300 python tools/bench_dec_pair.py 4096 2>&1 | grep -v amdgpu.ids > $O/dec_pair_timeline.txt
python3 bench.py > $O/bench_final.json 2> $O/bench_final.err
tail -c 600 $O/bench_final.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o dec -- python3 $R/tools/prof_decode.py > /tmp/prof_dec.log 2>&1
find /tmp/prof_dec -name "*kernel_stats.csv" -exec cp {} $O/decode_kernel_stats.csv \;
ls -la $O
