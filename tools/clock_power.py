#!/usr/bin/env python3
"""GPU box: shader clock and socket power while a bench runs.  Samples the amdgpu sysfs files (pp_dpm_sclk: the level marked '*',
hwmon power1_average / power1_input, gpu_busy_percent) every 20 ms from a thread while the command runs as a child process.
usage: clock_power.py <out.json> -- <command...>"""
import glob, json, os, subprocess, sys, threading, time


def find():
    devs = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(d + "/pp_dpm_sclk"):
            devs.append(d)
    return devs


def rd(p):
    try:
        return open(p).read()
    except Exception:
        return ""


def sample(d, hw):
    clk = None
    for ln in rd(d + "/pp_dpm_sclk").splitlines():
        if ln.strip().endswith("*"):
            clk = int("".join(c for c in ln.split(":")[1] if c.isdigit()))
    pw = None
    for f in ("power1_input", "power1_average"):
        for h in hw:
            s = rd(h + "/" + f).strip()
            if s.isdigit():
                pw = int(s) / 1e6
                break
        if pw is not None:
            break
    busy = rd(d + "/gpu_busy_percent").strip()
    return clk, pw, int(busy) if busy.isdigit() else None


def main():
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    devs = find()
    info = {"devices": devs}
    if not devs:
        info["error"] = "no amdgpu sysfs device with pp_dpm_sclk"
    hws = {d: glob.glob(d + "/hwmon/hwmon*") for d in devs}
    allsamples, stop = {d: [] for d in devs}, threading.Event()

    def loop():
        t0 = time.perf_counter()
        while not stop.is_set():
            for d in devs:
                allsamples[d].append((round(time.perf_counter() - t0, 3),) + sample(d, hws[d]))
            time.sleep(0.02)
    th = threading.Thread(target=loop); th.start()
    t0 = time.perf_counter()
    rc = subprocess.call(cmd)
    info["cmd_s"] = time.perf_counter() - t0
    stop.set(); th.join()
    info["rc"] = rc
    # the box shows every GPU of the node in sysfs; the one this job runs on is the one whose clock went up
    d = max(devs, key=lambda x: sum((s[1] or 0) for s in allsamples[x])) if devs else None
    samples = allsamples[d] if d else []
    info["device"] = d
    info["pp_dpm_sclk"] = rd(d + "/pp_dpm_sclk") if d else ""
    info["power_cap_W"] = [int(rd(h + "/power1_cap").strip() or 0) / 1e6 for h in hws.get(d, [])]
    info["n_samples"] = len(samples)
    info["samples"] = samples
    busy = [s for s in samples if (s[1] or 0) >= 1000]
    if busy:
        ck = sorted(s[1] for s in busy if s[1]); pw = sorted(s[2] for s in busy if s[2])
        info["while_busy"] = {"n": len(busy), "sclk_MHz": {"min": ck[0], "median": ck[len(ck) // 2], "max": ck[-1]} if ck else None,
                              "power_W": {"min": pw[0], "median": pw[len(pw) // 2], "max": pw[-1]} if pw else None}
    json.dump(info, open(out, "w"))
    print(json.dumps({k: v for k, v in info.items() if k != "samples"})[:1500])
    sys.exit(rc)


if __name__ == "__main__":
    main()
