"""gaussianBlur(sigma) of Image(u8) planes of several heights under rocprofv3: does a kernel's time scale with the rows, or is part of it fixed?
usage: rocprofv3 --kernel-trace -d <dir> -o r -- python tools/exp_sizes.py [sigma]; then python tools/exp_sizes.py --read <db>"""
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    import re, sqlite3
    c = sqlite3.connect(sys.argv[2])
    for n, g, cnt, avg, mn in c.execute(
            "select s.kernel_name, d.grid_size_x, count(*), avg(d.end - d.start), min(d.end - d.start) from rocpd_kernel_dispatch d "
            "join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like '%zg%' group by s.kernel_name, d.grid_size_x order by 1, 2"):
        print(f"{re.sub(r'[(].*', '', n)[:60]:60s} grid {g:9d} calls {cnt:3d} mean {avg / 1e3:8.2f} us min {mn / 1e3:8.2f} us")
    sys.exit(0)
sys.path.insert(0, ".")
import torch
import zignal_amd as zg

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.25
for rows in (512, 1024, 2048, 4096, 8192, 16384):
    s = zg.Image(torch.randint(0, 256, (rows, 4096), dtype=torch.uint8, device="cuda"))
    d = zg.Image(torch.empty_like(s.data))
    for _ in range(12):
        s.gaussian_blur(sigma, out=d)
    torch.cuda.synchronize()
