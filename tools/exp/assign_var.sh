for v in 1 2 3 4; do echo U $v; CLX_MS_DENSE_U=$v python tools/bench_stream.py ${1:-8192} 2>/dev/null | grep "ms_assign " | cut -c1-70; done
