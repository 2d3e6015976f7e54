#!/bin/bash
# Counter passes + kernel trace for one bench.py workload on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/run_pmc.sh [workload] [pass ...]'      passes: valu itype wait fetch write ktrace
# Each --pmc group is its own run (MI355X_MICROARCH.md: separate passes; never combined with sys/hip tracing).
# Raw CSVs land under gpurun_out/pmc_<workload>_<pass>/ and gpurun_out/ktrace_<workload>/; the summaries that get committed
# under profiles/ are written to gpurun_out/summaries/ (tools/pmc_summary.py, tools/rocprof_summary.py).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out
WL=${1:-merkle4_digests}
shift || true
PASSES=${*:-valu itype wait fetch write ktrace}
mkdir -p "$O/summaries"
EXTRA=""
TAG=$WL
# LOG2N=12 bash tools/run_pmc.sh merkle4_digests valu  -> the lane-group kernel of small batches (k_merkle4_coop<8>)
if [ -n "${LOG2N:-}" ]; then EXTRA="--log2n $LOG2N"; TAG=${WL}_$LOG2N; fi
[ "$WL" = tree ] && EXTRA="$EXTRA --no-check"  # a tree step is 12 launches: nothing but full builds may be in the trace (pmc_summary --per-step)
cd /tmp && export TMPDIR=/tmp
counters() {
    case $1 in
        valu)  echo SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE ;;
        itype) echo SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM ;;
        wait)  echo SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM ;;
        fetch) echo FETCH_SIZE ;;
        write) echo WRITE_SIZE ;;
        # stall attribution
        stall)  echo SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES ;;
        ifetch) echo SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_WAVES ;;
        icache) echo SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL ;;
        dcache) echo SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQC_TC_DATA_READ_REQ ;;
        tcp)    echo TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum ;;  # request rate of the gather (round 5)
        thread) echo SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE ;;
    esac
}
for p in $PASSES; do
    if [ "$p" = ktrace ]; then
        rm -rf "$O/ktrace_$TAG"
        timeout 300 rocprofv3 --kernel-trace --stats -d "$O/ktrace_$TAG" -o kt -- \
            python "$ROOT/bench.py" --workload "$WL" --no-cpu-baseline $EXTRA > "$O/ktrace_$TAG.log" 2>&1
    else
        rm -rf "$O/pmc_${TAG}_$p"
        timeout 300 rocprofv3 --pmc $(counters $p) --kernel-trace --output-format csv -d "$O/pmc_${TAG}_$p" -o pmc -- \
            python "$ROOT/bench.py" --workload "$WL" --steps 5 --warmup 1 --no-cpu-baseline $EXTRA > "$O/pmc_${TAG}_$p.log" 2>&1
    fi
    echo "pass $p rc=$?"
done
cd "$ROOT"
U20=1048576
case $WL in  # dominant kernel, permutations per launch, algorithmic bytes per permutation
    merkle4_digests) K=k_merkle4; U=$U20; B=160 ;;
    sponge42) K=k_sponge; U=$((12 * U20)); B=125.3333 ;;
    openings) K=k_merkle4_path; U=$((12 * U20)); B=102.3333 ;;  # (32 leaf + 12 x 96 siblings + 12 position bytes + 32 root) / 12
    encrypt) K=k_crypt; U=$((2 * U20)); B=128 ;;
    extract) K=k_merkle4_openings; U=$((12 * U20)); B=198.6667 ;;  # per (opening, level): 96 read + 96 written + 1 position byte, + (64 + 4) / 12 for the leaf and the index
    tree) K=k_merkle4; U=5592405; B=96.0000057; PS=--per-step; NAME=tree ;;
    *) K=k_merkle4; U=$U20; B=96 ;;
esac
PS=${PS:-}; NAME=${NAME:-$K}
if [ -n "${LOG2N:-}" ] && [ "$WL" = merkle4_digests ] && [ "$LOG2N" -le 13 ]; then K=k_merkle4_coop; U=$((1 << LOG2N)); NAME=k_merkle4_coop8; fi
dirs=""; for d in "$O"/pmc_${TAG}_[a-z]*; do [ -f "$d/pmc_counter_collection.csv" ] && dirs="$dirs $d"; done
[ -n "$dirs" ] && python tools/pmc_summary.py $K $U --bytes-per-unit $B $PS $dirs > "$O/summaries/pmc_$NAME.txt" 3> "$O/summaries/pmc_$NAME.json"
db=$(find "$O/ktrace_$TAG" -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" "bench.py --workload $WL" > "$O/summaries/ktrace_$TAG.txt"
ls "$O/summaries"
