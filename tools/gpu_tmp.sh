cd $GRAFT_REPO_ROOT
python tools/layer_profile.py 1 pspnet50 > gpurun_out/r04_layers_b1.txt 2>&1
python tools/layer_profile.py 2 pspnet50 > gpurun_out/r04_layers_b2_base.txt 2>&1
tail -16 gpurun_out/r04_layers_b1.txt
