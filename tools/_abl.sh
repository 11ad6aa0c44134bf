for a in 0 1 2 3 4; do echo "ablate $a"; LZX_ABLATE=$a LZX_TRACE_SPMV=1 timeout -k 10 120 python tools/perf_probe.py c3 @one 2>&1 | grep trace; done
