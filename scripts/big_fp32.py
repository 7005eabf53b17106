import sys, time
sys.path.insert(0, "/root/repo")
from dense_linear_app_amd import chameleon as ch
ch.CHAMELEON_Init(1, 1)
N, B = 131072, 1024
d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealFloat, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
t0 = time.perf_counter(); info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d); dt = time.perf_counter() - t0
print(f"N={N} B={B} f32 info={info} {dt*1e3:.1f} ms {N**3/3/dt/1e12:.2f} TF/s", flush=True)
print("residual", ch.residual_plgsy(d, float(N), 42), flush=True)
