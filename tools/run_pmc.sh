#!/bin/bash
# Counter passes + kernel trace for one bench.py workload on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/run_pmc.sh [workload] [pass ...]'      passes: valu fetch write wait cache itype ktrace
# Each --pmc group is its own run (MI355X_MICROARCH.md: separate passes; never combined with sys/hip tracing).
# Outputs land under gpurun_out/pmc_<pass>/ and gpurun_out/ktrace/; tools/pmc_summary.py and
# tools/rocprof_summary.py turn them into the text files committed under profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
WL=${1:-merkle4_digests}
shift || true
PASSES=${*:-valu fetch write wait cache itype ktrace}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
counters() {
    case $1 in
        valu)  echo SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE ;;
        fetch) echo FETCH_SIZE ;;
        write) echo WRITE_SIZE ;;
        wait)  echo SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_SALU ;;
        cache) echo SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES ;;
        itype) echo SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_IFETCH SQ_INST_LEVEL_SMEM ;;
    esac
}
for p in $PASSES; do
    if [ "$p" = ktrace ]; then
        rm -rf "$OUT/ktrace"
        timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/ktrace" -o kt -- \
            python "$ROOT/bench.py" --workload "$WL" --no-cpu-baseline > "$OUT/ktrace.log" 2>&1
    else
        rm -rf "$OUT/pmc_$p"
        timeout 300 rocprofv3 --pmc $(counters $p) --kernel-trace --output-format csv -d "$OUT/pmc_$p" -o pmc -- \
            python "$ROOT/bench.py" --workload "$WL" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$p.log" 2>&1
    fi
    echo "pass $p rc=$?"
done
