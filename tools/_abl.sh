for a in 0 6 7 8 1 0; do echo "ablate $a"; LZX_ABLATE=$a LZX_TRACE_SPMV=1 timeout -k 10 120 python tools/perf_probe.py c3 @one5 2>&1 | grep trace | head -1; done
