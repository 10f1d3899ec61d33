"""Power and shader clock of the GPU while a command runs (sysfs hwmon, read-only):  python tools/power_sampler.py <out.txt> -- <command ...>
Samples every ~20 ms; prints mean / p10 / p90 of the package power and of the shader clock over the samples taken while the command ran,
and (when the command prints bench.py's JSON line) nothing else of it.  Used for DESIGN 5: is a C3 step power-limited?"""
import glob, os, subprocess, sys, time, threading

def my_card():
    """/sys/class/drm/cardN of the GPU this process's HIP device 0 is (several cards are visible in sysfs on a shared node)"""
    try:
        import torch
        pr = torch.cuda.get_device_properties(0)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except Exception as e:
        print("no PCI id of device 0:", e)
        return "/sys/class/drm/card*"
    for c in glob.glob("/sys/class/drm/card*"):
        if os.path.basename(os.path.realpath(os.path.join(c, "device"))).startswith(want):
            print("device 0 =", c, want)
            return c
    print("no card matches", want)
    return "/sys/class/drm/card*"

def sources():
    out = {}
    card = my_card()
    for h in glob.glob(card + "/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp2_input"):
            f = os.path.join(h, name)
            if os.path.exists(f):
                out.setdefault(name, f)
    for d in glob.glob(card + "/device"):
        for name in ("gpu_busy_percent", "pp_dpm_sclk"):
            f = os.path.join(d, name)
            if os.path.exists(f):
                out.setdefault(name, f)
    return out

def read(f):
    try:
        return open(f).read().strip()
    except OSError:
        return ""

def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    src = sources()
    samples = []
    stop = threading.Event()
    def loop():
        while not stop.is_set():
            row = {"t": time.time()}
            for k, f in src.items():
                v = read(f)
                if k == "pp_dpm_sclk":
                    cur = [l for l in v.splitlines() if l.endswith("*")]
                    row[k] = float(cur[0].split()[1].lower().replace("mhz", "")) if cur else float("nan")
                else:
                    try:
                        row[k] = float(v)
                    except ValueError:
                        row[k] = float("nan")
            samples.append(row)
            time.sleep(0.02)
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    t0 = time.time()
    rc = subprocess.call(cmd)
    t1 = time.time()
    stop.set(); th.join()
    with open(out_path, "w") as fo:
        fo.write(f"# sources: {src}\n# command: {' '.join(cmd)} (rc {rc}, {t1 - t0:.1f} s, {len(samples)} samples)\n")
        keys = [k for k in src]
        fo.write("t " + " ".join(keys) + "\n")
        for r in samples:
            fo.write(f"{r['t'] - t0:.3f} " + " ".join(f"{r.get(k, float('nan')):.0f}" for k in keys) + "\n")
    def stat(k, scale, lo_t, hi_t):
        v = sorted(r[k] * scale for r in samples if k in r and r[k] == r[k] and lo_t <= r["t"] - t0 <= hi_t)
        if not v:
            return "n/a"
        return f"mean {sum(v) / len(v):.0f}, p10 {v[len(v) // 10]:.0f}, p50 {v[len(v) // 2]:.0f}, p90 {v[len(v) * 9 // 10]:.0f}, max {v[-1]:.0f} ({len(v)} samples)"
    dur = t1 - t0
    for k, scale, unit in (("power1_average", 1e-6, "W"), ("power1_input", 1e-6, "W"), ("freq1_input", 1e-6, "MHz sclk"), ("pp_dpm_sclk", 1.0, "MHz sclk (dpm)"),
                           ("gpu_busy_percent", 1.0, "% busy"), ("temp2_input", 1e-3, "C"), ("power1_cap", 1e-6, "W cap")):
        if k in src:
            print(f"{k} [{unit}] whole run: {stat(k, scale, 0, dur)}")
            print(f"{k} [{unit}] last 40 %: {stat(k, scale, 0.6 * dur, dur)}")
    sys.exit(rc)

main()
