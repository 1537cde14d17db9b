import sys, time, torch
sys.path.insert(0, "/root/repo")
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
osd = make_synthetic_original_state_dict("vitl", 0)
models = []
for i in range(2):
    _, m = make_depthanythingv2_dpt_from_original_state_dict(osd)
    models.append(m.to("cuda", torch.bfloat16))
x = torch.randn(32, 3, 504, 504, device="cuda", dtype=torch.bfloat16)
def timeit(fn, steps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps
with torch.inference_mode():
    t1 = timeit(lambda: models[0](x))
    print("1 stream  B=32: %.2f ms  %.1f maps/s" % (t1 * 1e3, 32 / t1))
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    xs = [x[:16].contiguous(), x[16:].contiguous()]
    def two():
        for i in range(2):
            with torch.cuda.stream(s[i]):
                models[i](xs[i])
    t2 = timeit(two)
    print("2 streams B=16+16: %.2f ms  %.1f maps/s" % (t2 * 1e3, 32 / t2))
    t3 = timeit(lambda: models[0](xs[0]))
    print("1 stream  B=16: %.2f ms  %.1f maps/s" % (t3 * 1e3, 16 / t3))
