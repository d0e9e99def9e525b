"""sdod.amd -- MI355X (gfx950) host side of the txt2img hot path: ctypes bindings to the C-ABI
kernel/engine library (`lib/libsdod.so`) plus the Python sampler loop (north_star: "Python host
code on PyTorch-ROCm drives the PLMS/DPM sampler loop").  PyTorch is used only for device memory,
streams and torch.distributed."""
