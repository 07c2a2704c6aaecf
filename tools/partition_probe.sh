# read-only probe of the box's partition modes (round 6, review item 4a): can one physical MI355X show up as two logical devices?
mkdir -p gpurun_out/r6
{
echo "== rocm-smi --showcomputepartition"; rocm-smi --showcomputepartition 2>&1 | head -20
echo "== rocm-smi --showmemorypartition"; rocm-smi --showmemorypartition 2>&1 | head -20
echo "== amd-smi static partition"; amd-smi static --partition 2>&1 | head -40
echo "== sysfs"; for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_compute_partition /sys/class/drm/card*/device/current_memory_partition; do echo "$f: $(cat $f 2>&1)"; done
echo "== devices"; python -c "import torch; print(torch.cuda.device_count()); print(torch.cuda.get_device_properties(0))"
echo "== id"; id; ls -l /dev/kfd /dev/dri 2>&1 | head
} > gpurun_out/r6/partition_probe.txt 2>&1
