export PYTHONUNBUFFERED=1
SDOD_ATTN_QT1=1 timeout -k 10 300 python tools/unet_profile.py unet --top 80 > gpurun_out/unet_prof8.txt 2>&1; grep -E "attn_|launches" gpurun_out/unet_prof8.txt
