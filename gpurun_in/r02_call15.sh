#!/bin/bash
# round 2, GPU call 15: NUMA placement probe for the pageable host path
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r02o; mkdir -p $O
{
echo "== numa nodes"; ls /sys/devices/system/node/ | grep node; for n in /sys/devices/system/node/node*; do echo "$n cpulist $(cat $n/cpulist)"; done
echo "== gpu pci"; python - <<'PY'
import torch, glob, os
p = torch.cuda.get_device_properties(0)
print(p.name, getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None), getattr(p, "pci_domain_id", None))
for d in glob.glob("/sys/class/drm/card*/device"):
    try:
        print(d, os.path.realpath(d), "numa_node", open(d + "/numa_node").read().strip(), "local_cpulist", open(d + "/local_cpulist").read().strip(), "vendor", open(d + "/vendor").read().strip())
    except OSError as e:
        print(d, e)
print("affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
print("cpuset", open("/sys/fs/cgroup/cpuset.cpus.effective").read().strip() if os.path.exists("/sys/fs/cgroup/cpuset.cpus.effective") else None)
PY
which numactl taskset
echo "== default placement, 3 runs"
for i in 1 2 3; do python bench_tools/host_path_bench.py 2>/dev/null | grep "2^22"; done
for node in /sys/devices/system/node/node*; do
  cl=$(cat $node/cpulist); echo "== taskset -c $cl"
  taskset -c $cl python bench_tools/host_path_bench.py 2>/dev/null | grep "2^22"
done
} > $O/numa_probe.txt 2>&1
cat $O/numa_probe.txt
