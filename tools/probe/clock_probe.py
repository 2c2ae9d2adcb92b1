"""what the amdsmi python binding reports on this box while a GEMM loop runs (clock / power / temperature fields)"""
import threading, time, json, sys
import torch
import amdsmi
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
h = hs[0]
def snap():
    out = {}
    for name, fn in (('metrics', lambda: amdsmi.amdsmi_get_gpu_metrics_info(h)),
                     ('clock_gfx', lambda: amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)),
                     ('power', lambda: amdsmi.amdsmi_get_power_info(h)),
                     ('temp', lambda: amdsmi.amdsmi_get_temp_metric(h, amdsmi.AmdSmiTemperatureType.HOTSPOT, amdsmi.AmdSmiTemperatureMetric.CURRENT))):
        t0 = time.perf_counter()
        try:
            v = fn()
        except Exception as e:  # noqa
            v = 'ERR ' + repr(e)[:100]
        out[name] = (round((time.perf_counter() - t0) * 1e3, 2), v)
    return out
s = snap()
for k, (ms, v) in s.items():
    if isinstance(v, dict):
        v = {kk: vv for kk, vv in v.items() if any(t in kk for t in ('clk', 'power', 'temp', 'throttle', 'activity'))}
    print(k, ms, 'ms', json.dumps(v, default=str)[:1500])
a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
stop = False
rows = []
def loop():
    while not stop:
        m = amdsmi.amdsmi_get_gpu_metrics_info(h)
        rows.append((m.get('current_gfxclks', [None])[:8], m.get('average_socket_power'), m.get('current_socket_power'), m.get('temperature_hotspot')))
        time.sleep(0.1)
th = threading.Thread(target=loop); th.start()
for _ in range(3000):
    a @ b
torch.cuda.synchronize()
stop = True; th.join()
for r in rows[::3]:
    print(r)
