mkdir -p gpurun_out/r06s
timeout 600 python tools/fill_load_probe.py 24 fill acc > gpurun_out/r06s/fill_load_soak_acc.json 2> gpurun_out/r06s/acc.err; tail -c 600 gpurun_out/r06s/fill_load_soak_acc.json; tail -2 gpurun_out/r06s/acc.err
timeout 600 python tools/fill_load_probe.py 48 fill jrk > gpurun_out/r06s/fill_load_soak_jrk.json 2> gpurun_out/r06s/jrk.err; tail -c 600 gpurun_out/r06s/fill_load_soak_jrk.json; tail -2 gpurun_out/r06s/jrk.err
