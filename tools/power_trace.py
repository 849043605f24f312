"""Clock / power of the GPU while a command runs (sysfs, ~5 ms period): does the training step run against the power cap?
    python tools/power_trace.py -- python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-sample --no-secondary
Prints min / median / max of the shader clock and the average socket power over the busy part of the run."""
import glob, os, subprocess, sys, threading, time

def find():
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(card + "/pp_dpm_sclk"):
            hw = glob.glob(card + "/hwmon/hwmon*")
            return card, (hw[0] if hw else None)
    return None, None

def read(path):
    try:
        return open(path).read()
    except OSError:
        return ""

def sclk(card):
    for line in read(card + "/pp_dpm_sclk").splitlines():
        if line.strip().endswith("*"):
            return int("".join(c for c in line.split(":")[1] if c.isdigit()))
    return 0

def main():
    cmd = sys.argv[sys.argv.index("--") + 1:]
    card, hw = find()
    print("card", card, "hwmon", hw)
    if hw:
        for f in ("power1_cap", "power1_cap_max", "power1_average", "power1_input", "freq1_input", "temp1_input"):
            v = read(hw + "/" + f).strip()
            if v:
                print(" ", f, v)
    samples, stop = [], [False]
    def loop():
        while not stop[0]:
            t = time.time()
            p = read(hw + "/power1_average").strip() or read(hw + "/power1_input").strip() if hw else ""
            f = read(hw + "/freq1_input").strip() if hw else ""
            samples.append((t, sclk(card), int(p) / 1e6 if p else 0.0, int(f) / 1e6 if f else 0.0))
            time.sleep(0.005)
    th = threading.Thread(target=loop); th.start()
    out = subprocess.run(cmd, capture_output=True, text=True)
    stop[0] = True; th.join()
    print(out.stdout.strip().splitlines()[-1][:200] if out.stdout.strip() else out.stderr[-300:])
    busy = [s for s in samples if s[2] > 0.5 * max(x[2] for x in samples)] or samples
    for name, idx in (("sclk MHz (pp_dpm_sclk)", 1), ("power W", 2), ("freq1_input MHz", 3)):
        v = sorted(s[idx] for s in busy)
        print("%-24s n %5d  min %8.1f  p10 %8.1f  median %8.1f  p90 %8.1f  max %8.1f" % (name, len(v), v[0], v[len(v) // 10], v[len(v) // 2], v[len(v) * 9 // 10], v[-1]))

if __name__ == "__main__":
    main()
