import os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from elmkernels_amd import state as st
cols = 1_000_000
pads = [0, 1, 3, 17, 64, 255, 1024, 0, 0]
keep = []
for i, mb in enumerate(pads):
    if mb:
        keep.append(torch.empty(mb * 1024 * 1024, dtype=torch.uint8, device="cuda"))
    D, _ = bench.build_state(cols, 0, "A", 0x5EEDE1A0)
    for _ in range(6):
        D.restore_fields(); st.timestep7(D, 1800.0)
    for _ in range(4):
        D.restore_fields(); st.timestep7_fused(D, 1800.0)
    D.restore_fields()
    ms, tot = D.profile_timestep7_fused(1800.0, 10)
    ptr = D.device_ptr("t_soisno") if hasattr(D, "device_ptr") else 0
    print(f"alloc {i} (pad {mb} MB before): " + " ".join(f"{n}={m:.3f}" for n, m in zip(st.KERNEL_NAMES_FUSED, ms)) + f" total {tot:.3f} ptr {ptr:#x}", flush=True)
    D.close()
