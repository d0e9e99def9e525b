export PYTHONUNBUFFERED=1
timeout -k 10 300 python tools/unet_profile.py vae --top 40 > gpurun_out/vae_prof2.txt 2>&1; head -44 gpurun_out/vae_prof2.txt
