F='amdgpu.ids|socket.cpp|Gloo'
SIMHAND_GLOO_ASYNC_BUCKETS=1 PROBE_COLD=1 timeout 400 python scripts/syncbn_repeat_probe.py 300 2>&1 | grep -Ev "$F" | tail -14
