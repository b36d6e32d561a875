python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch
print('torch alone', torch.cuda.is_available(), torch.cuda.device_count())
import sys; sys.path.insert(0,'.')
from pybader_amd import _lib
c = _lib.Context(0); print('ctx after torch ok')
import numpy as np
t = torch.zeros(4, device='cuda:0'); print(t.sum().item())
PY
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0,'.')
from pybader_amd import _lib
c = _lib.Context(0); print('ctx first ok')
import torch
print('torch after lib', torch.cuda.is_available(), torch.cuda.device_count())
PY
env | grep -i -E "HIP|ROCR|CUDA|HSA" 
