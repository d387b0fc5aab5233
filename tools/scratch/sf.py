import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import siggen, sdr_server_amd as xl
FS=2016000
x = siggen.xs_u8(1, 262144)
taps = xl.create_low_pass_filter(1.0, FS, 24000, 48000)[1]
for variant in ("native","optimized","native"):
    f = xl.XlatingFilter(42, taps, -12000, FS, 262144)
    for _ in range(20): f.process(variant, "cu8", "cf32", x)
    t0=time.perf_counter()
    for _ in range(200): f.process(variant, "cu8", "cf32", x)
    print(variant, len(taps), (time.perf_counter()-t0)/200*1e6, "us")
    f.close()
