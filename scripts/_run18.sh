LOW=gru D=128 bash scripts/profile_refbatch_train.sh 2>&1 | grep -v "rocprofv3\|amdgpu.ids" | cut -c1-160 | head -26
LOW=tcn D=256 bash scripts/profile_refbatch_train.sh 2>&1 | grep -v "rocprofv3\|amdgpu.ids" | cut -c1-160 | head -26
