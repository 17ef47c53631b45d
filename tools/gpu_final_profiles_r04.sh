# Round-4 end-of-round evidence (run on the GPU box through gpurun): bench lines, rocprofv3 kernel-trace summaries of the same
# commands, PMC traffic, the size sweep.  ROUND=r04 bash tools/gpu_final_profiles_r04.sh
ROUND=${ROUND:-r04}
export ROUND
bash tools/gpu_final_profiles.sh
