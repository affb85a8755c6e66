"""Recover the numeric series from the reference's committed matplotlib vector PDFs
(``Assets/ReportResults/<run>/evolutions/evolution_<i>.pdf``).  These figures are the only
numeric outputs of the reference's CasADi/IPOPT path that exist anywhere (the reference has no
tests or fixtures), so they serve as *soft* goldens for the closed loop (IPOPT tol=1e-5).

A figure holds one Flate stream with the page content.  Data polylines are runs of
``x y m`` / ``x y l`` with >= 9 points; tick marks are 2-point paths closed by ``B`` and
followed by a ``[(label)] TJ``; the minus glyph of negative labels is not recoverable, so signs
are inferred by requiring tick values to be an arithmetic progression increasing with
position.  A least-squares ``value = k * pt + b`` per axis maps points to data units.

Run in the build container only (needs /root/reference); output: pdf_series.npz next to it.
"""
import re
import sys
import zlib

import numpy as np

NUM = r"-?\d+\.?\d*(?:e-?\d+)?"


def _page_stream(path):
    d = open(path, "rb").read()
    best = b""
    for m in re.finditer(rb"stream\r?\n(.*?)endstream", d, re.S):
        try:
            t = zlib.decompress(m.group(1))
        except zlib.error:
            continue
        if len(t) > len(best):
            best = t
    return best.decode("latin1")


def _ticks(txt):
    xt, yt = [], []
    pat = re.compile(
        rf"({NUM}) ({NUM}) m\n({NUM}) ({NUM}) l\n\nB\n.*?\[\s*\(([^)]*)\)\s*\] TJ", re.S)
    for m in pat.finditer(txt):
        x1, y1, x2, y2 = (float(m.group(i)) for i in range(1, 5))
        lab = m.group(5)
        digits = re.sub(r"[^0-9.]", "", lab)
        if not digits:
            continue
        val = float(digits)
        if abs(x1 - x2) < 1e-9:
            xt.append((x1, val))
        elif abs(y1 - y2) < 1e-9:
            yt.append((y1, val))
    return xt, yt


def _calibrate(ticks):
    """ticks: (position, |value|).  Choose signs making values an increasing AP."""
    ticks = sorted(set(ticks))
    pos = np.array([t[0] for t in ticks])
    mag = np.array([t[1] for t in ticks])
    best = None
    n = len(ticks)
    # negative labels form a prefix (values increase with position)
    for nneg in range(n + 1):
        val = mag.copy()
        val[:nneg] *= -1.0
        if np.any(np.diff(val) <= 0):
            continue
        k, b = np.polyfit(pos, val, 1)
        err = np.max(np.abs(k * pos + b - val))
        if best is None or err < best[0]:
            best = (err, k, b)
    if best is None:
        raise ValueError("no consistent tick signs")
    return best[1], best[2], best[0]


def _polylines(txt, min_pts=9):
    out = []
    for m in re.finditer(rf"(?:{NUM} {NUM} [ml]\n){{{min_pts},}}", txt):
        pts = np.array([[float(a), float(b)] for a, b in re.findall(rf"({NUM}) ({NUM}) [ml]", m.group(0))])
        out.append(pts)
    return out


def extract(path):
    txt = _page_stream(path)
    xt, yt = _ticks(txt)
    kx, bx, ex = _calibrate(xt)
    ky, by, ey = _calibrate(yt)
    series = []
    for pts in _polylines(txt):
        series.append(np.stack([kx * pts[:, 0] + bx, ky * pts[:, 1] + by], axis=1))
    return series, (ex, ey)


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Assets/ReportResults"
    out = {}
    for run in ("Simulation1Circles", "Simulation1CirclesDelta", "Simulation1"):
        for i in range(4):
            series, err = extract(f"{root}/{run}/evolutions/evolution_{i}.pdf")
            for j, s in enumerate(series):
                out[f"{run}/ev{i}/s{j}"] = s
            print(run, i, [s.shape for s in series], "tick fit err", err)
    np.savez_compressed(__file__.replace("extract_pdf_series.py", "pdf_series.npz"), **out)
