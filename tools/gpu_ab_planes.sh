set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider -k "planes or winograd_conv3x3" > gpurun_out/r04_planes_ops.txt 2>&1; rc=$?; tail -5 gpurun_out/r04_planes_ops.txt
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
timeout -k 10 300 python tools/layer_profile.py 2 pspnet50 hip_plane_operands > gpurun_out/r04_layers_planes_$i.txt 2>&1 && tail -16 gpurun_out/r04_layers_planes_$i.txt
timeout -k 10 300 python tools/layer_profile.py 2 pspnet50 > gpurun_out/r04_layers_noplanes_$i.txt 2>&1 && tail -16 gpurun_out/r04_layers_noplanes_$i.txt
done
