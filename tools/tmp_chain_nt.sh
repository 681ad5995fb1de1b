set -e
R=65536 timeout 300 python tools/gpu_chain_probe.py 2>&1 | grep -i "online\|momentum\|bwd"
