set -x
timeout -k 10 300 python tools/aten_by_site.py > gpurun_out/r3_aten_sites.txt 2>&1 && timeout -k 10 300 python tools/cast_sites.py > gpurun_out/r3_cast_sites.txt 2>&1
echo rc=$?
