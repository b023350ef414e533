#!/bin/bash
# usage: [CFD_GIT_HEAD=$(git rev-parse HEAD)] scripts/collect_profiles.sh <tag> <bench.py args...>
# Four rocprofv3 runs of the SAME bench.py command (run on the GPU box via gpurun):
#   1. --kernel-trace --stats              -> per-kernel durations
#   2. --pmc FETCH_SIZE                    -> HBM-side read traffic   (own pass: TCC slots)
#   3. --pmc WRITE_SIZE                    -> HBM-side write traffic  (own pass)
#   4. --pmc MFMA busy / GRBM_GUI_ACTIVE   -> matrix-core utilisation and effective clock
# Raw output goes to gpurun_out/prof_<tag>/{stats,fetch,write,mfma}; scripts/profile_summary.py condenses it
# into profiles/.
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
# identity of the tree these passes run on (the box has no .git): digest of the kernel sources + host modules, compared by
# bench.py with the running tree when it quotes these counters (roofline.traffic_source.same_sources_as_this_run)
python3 -c "import sys; sys.path.insert(0, '$R'); import bench, json; json.dump({'csrc_sha16': bench.csrc_sha16(), 'git_head': '${CFD_GIT_HEAD:-}'}, open('$O/source_stamp.json', 'w'))" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/bench_stats.json 2> $O/stats.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/bench_fetch.json 2> $O/fetch.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/bench_write.json 2> $O/write.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/bench_mfma.json 2> $O/mfma.err || exit 1
