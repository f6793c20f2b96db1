python tools/trace_bench.py --workload mesh_1m --traversal 1 3 2 > gpurun_out/tb_mesh.txt 2>&1
python tools/trace_bench.py --workload blob_870k --traversal 1 3 > gpurun_out/tb_blob.txt 2>&1
python tools/option_sweep.py --workload mesh_1m > gpurun_out/sw_mesh.txt 2>&1
python tools/option_sweep.py --workload blob_870k > gpurun_out/sw_blob.txt 2>&1
