timeout 60 tools/micro/vmm_probe
