#!/bin/bash
# One measurement pass over everything profiles/ quotes for the current build; run on the GPU box:
#   gpurun --timeout 1200 -- 'bash tools/final_pass.sh r3f'
# writes gpurun_out/<tag>/ (and gpurun_out/prof_<tag>*/); tools/collect_profiles.py copies the summaries into profiles/.
TAG=${1:-r3f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done"; tail -c 600 $OUT/bench.json
bash tools/profile_gpu.sh $TAG > $OUT/profile.log 2>&1 || exit 2
FLAGS="--table-lds" bash tools/profile_gpu.sh ${TAG}_tablds > $OUT/profile_tablds.log 2>&1 || exit 3
echo "profiles done"
python3 tools/bench_configs.py > $OUT/other_configs.json 2> $OUT/other_configs.err || exit 4
python3 tools/bench_mesh.py > $OUT/mesh_scenes.json 2>/dev/null || exit 5
python3 tools/mesh_stats.py > $OUT/mesh_stats.json 2>/dev/null || exit 6
python3 tools/fast_mode_report.py > $OUT/fast_mode.json 2>/dev/null || exit 7
python3 tools/host_overhead.py --streams 1 > $OUT/host_overhead_single_stream.json 2>/dev/null || exit 8
python3 tools/host_overhead.py --streams 2 > $OUT/host_overhead_two_streams.json 2>/dev/null || exit 9
python3 tools/bench_multi.py > $OUT/multi_one_gpu_rehearsal.json 2>/dev/null || exit 10
python3 tools/first_launch.py --build-wall-s ${BUILD_WALL_S:-26.5} --build-cpu-s ${BUILD_CPU_S:-95} > $OUT/build_and_first_launch.json 2>/dev/null || exit 15
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_flags.json 2>/dev/null || exit 16
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --share-gpu > $OUT/bench_n2_gloo_one_gpu.json 2>/dev/null || exit 17
echo "configs done"
bash tools/ablate_pmc.sh > $OUT/ablation_pmc.txt 2>&1 || exit 11
OUT=/tmp/pmc_classes_$TAG bash tools/pmc_classes.sh > $OUT/pmc_classes.txt 2>&1 || exit 18
RT_ENGINE_LIB=$PWD/ray-tracer-engine_amd/csrc/librt_engine_tuning.so python3 tools/mesh_ablate.py > $OUT/mesh_ablate.json 2>/dev/null || exit 12
echo "ablation done"
python3 tools/tile_order_experiment.py > $OUT/tile_order.json 2>/dev/null || exit 13
python3 bench.py --no-cpu-baseline --tile-order 0 > $OUT/bench_tile_order0.json 2>/dev/null || exit 14
echo "launch order done"
