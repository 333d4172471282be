"""Shader clock / socket power / junction temperature of ONE GPU, sampled from a side thread — MEASUREMENT INFRASTRUCTURE
(bench.py puts the averages of its timed region into every result line; tools/energy_table.py turns them into joules per launch).

Why: the step runs against the socket's power cap, boxes differ by +-3 % in the clock they sustain,
and a bench line that carries neither clock nor power cannot be compared with the line of another box or another round.

Source: the `amdsmi` Python package of the ROCm image, in-process (one ioctl-backed call per sample, ~0.2 ms), matched to the
HIP device through its PCI address; when amdsmi is missing or its calls fail, `rocm-smi --json` in a subprocess (slow: ~2 Hz).
Reads only; never changes a clock, a cap or a profile.  No sample => the fields are null, the bench line stays valid.

    python tools/gpu_telemetry.py            # prints what every source returns on this box (field names differ by release)
"""
import json
import subprocess
import threading
import time
from typing import Any, Dict, List, Optional


def _num(v: Any) -> Optional[float]:
    try:
        if v is None or isinstance(v, str) and not v.strip().replace(".", "", 1).replace("-", "", 1).isdigit():
            return None
        f = float(v)
        return f if f == f and abs(f) < 1e12 else None
    except (TypeError, ValueError):
        return None


class _AmdSmiSource:
    """one GPU through amdsmi; raises at construction when the package / the device is not usable"""

    name = "amdsmi"

    def __init__(self, pci_bdf: Optional[str], index: int) -> None:
        import amdsmi

        self.smi = amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
        if not handles:
            raise RuntimeError("amdsmi sees no GPU")
        self.handle = None
        if pci_bdf:
            for h in handles:
                try:
                    if str(amdsmi.amdsmi_get_gpu_device_bdf(h)).lower() == pci_bdf.lower():
                        self.handle = h
                except Exception:
                    pass
        if self.handle is None:
            self.handle = handles[index if index < len(handles) else 0]
        self.matched_by = "pci address" if pci_bdf and self.handle is not None and self._bdf() == pci_bdf.lower() else "index"
        self.cap_w = self._cap()
        if self.sample()["power_w"] is None and self.sample()["sclk_mhz"] is None:
            raise RuntimeError("amdsmi returns neither power nor clock for this GPU")

    def _bdf(self) -> str:
        try:
            return str(self.smi.amdsmi_get_gpu_device_bdf(self.handle)).lower()
        except Exception:
            return ""

    def _cap(self) -> Optional[float]:
        try:
            info = self.smi.amdsmi_get_power_cap_info(self.handle)
            v = _num(info.get("power_cap"))
            if v is None:
                return None
            return v / 1e6 if v > 1e5 else v  # (microwatts in the releases that follow the C API, watts in newer ones)
        except Exception:
            return None

    def sample(self) -> Dict[str, Optional[float]]:
        smi, h = self.smi, self.handle
        sclk = power = temp = None
        try:
            m = smi.amdsmi_get_gpu_metrics_info(h)
            clks = [c for c in (_num(c) for c in (m.get("current_gfxclks") or [])) if c and c < 60000]
            sclk = (sum(clks) / len(clks)) if clks else _num(m.get("current_gfxclk"))
            power = _num(m.get("current_socket_power")) or _num(m.get("average_socket_power"))
            temp = _num(m.get("temperature_hotspot"))
        except Exception:
            pass
        if power is None or power <= 0 or power > 5000:
            try:
                p = smi.amdsmi_get_power_info(h)
                power = _num(p.get("current_socket_power")) or _num(p.get("average_socket_power")) or _num(p.get("socket_power"))
            except Exception:
                power = None
        if sclk is None or sclk <= 0 or sclk > 60000:
            try:
                c = smi.amdsmi_get_clock_info(h, smi.AmdSmiClkType.GFX)
                sclk = _num(c.get("clk")) or _num(c.get("cur_clk"))
            except Exception:
                sclk = None
        if temp is None or temp <= 0 or temp > 200:
            try:
                temp = _num(smi.amdsmi_get_temp_metric(h, smi.AmdSmiTemperatureType.HOTSPOT, smi.AmdSmiTemperatureMetric.CURRENT))
            except Exception:
                temp = None
        return {"sclk_mhz": sclk, "power_w": power, "junction_c": temp}


class _RocmSmiSource:
    name = "rocm-smi"

    def __init__(self, pci_bdf: Optional[str], index: int) -> None:
        self.index = index
        self.matched_by = "index"
        self.cap_w = None
        try:
            out = subprocess.run(["rocm-smi", "-d", str(index), "--showmaxpower", "--json"], capture_output=True, text=True, timeout=20).stdout
            for card in json.loads(out[out.index("{"):]).values():
                for k, v in card.items():
                    if "max" in k.lower() and "power" in k.lower():
                        self.cap_w = _num(v)
        except Exception:
            pass
        if self.sample()["power_w"] is None:
            raise RuntimeError("rocm-smi returns no power figure")

    def sample(self) -> Dict[str, Optional[float]]:
        sclk = power = temp = None
        try:
            out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showpower", "--showclocks", "--showtemp", "--json"],
                                 capture_output=True, text=True, timeout=20).stdout
            for card in json.loads(out[out.index("{"):]).values():
                for k, v in card.items():
                    kl = k.lower()
                    if "sclk clock speed" in kl:
                        sclk = _num(str(v).strip("()").lower().replace("mhz", ""))
                    elif "power (w)" in kl and "max" not in kl:
                        power = _num(v)
                    elif "junction" in kl:
                        temp = _num(v)
        except Exception:
            pass
        return {"sclk_mhz": sclk, "power_w": power, "junction_c": temp}


def pci_address_of(device_index: int) -> Optional[str]:
    """'0000:05:00.0' of a HIP device as torch reports it (None when this torch build does not expose it)"""
    try:
        import torch

        p = torch.cuda.get_device_properties(device_index)
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except Exception:
        return None


class GpuTelemetry:
    """`with tel.window() as w: ...timed region...` -> w.summary() = averages over the samples that fell inside the window.
    The sampler thread runs from `start()` to `stop()` at `hz` (default 10 with amdsmi); windows only pick their samples."""

    def __init__(self, device_index: int = 0, hz: float = 10.0) -> None:
        self.source: Any = None
        self.error: Optional[str] = None
        bdf = pci_address_of(device_index)
        for cls in (_AmdSmiSource, _RocmSmiSource):
            try:
                self.source = cls(bdf, device_index)
                break
            except Exception as e:  # next source
                self.error = f"{cls.name}: {type(e).__name__}: {e}"[:200]
        self.hz = hz if (self.source is not None and self.source.name == "amdsmi") else min(hz, 2.0)
        self.samples: List[tuple] = []  # (t, sclk, power, temp)
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None

    @property
    def available(self) -> bool:
        return self.source is not None

    def start(self) -> "GpuTelemetry":
        if self.source is None or self._thread is not None:
            return self
        self._stop.clear()

        def loop() -> None:
            period = 1.0 / self.hz
            while not self._stop.is_set():
                t = time.perf_counter()
                s = self.source.sample()
                self.samples.append((t, s["sclk_mhz"], s["power_w"], s["junction_c"]))
                self._stop.wait(max(0.0, period - (time.perf_counter() - t)))

        self._thread = threading.Thread(target=loop, name="gpu-telemetry", daemon=True)
        self._thread.start()
        return self

    def stop(self) -> None:
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=5.0)
            self._thread = None

    def sample_now(self) -> Dict[str, Optional[float]]:
        return self.source.sample() if self.source is not None else {"sclk_mhz": None, "power_w": None, "junction_c": None}

    def summary(self, t0: float, t1: float) -> Dict[str, Any]:
        """averages over the samples taken in [t0, t1] (perf_counter times)"""
        rows = [r for r in self.samples if t0 <= r[0] <= t1]

        def avg(i: int) -> Optional[float]:
            v = [r[i] for r in rows if r[i] is not None]
            return round(sum(v) / len(v), 1) if v else None

        def lo_hi(i: int) -> Optional[list]:
            v = [r[i] for r in rows if r[i] is not None]
            return [round(min(v), 1), round(max(v), 1)] if v else None

        return {"sclk_mhz_avg": avg(1), "power_w_avg": avg(2), "junction_c_avg": avg(3), "sclk_mhz_range": lo_hi(1),
                "power_w_range": lo_hi(2), "power_cap_w": None if self.source is None else self.source.cap_w, "samples": len(rows),
                "window_s": round(t1 - t0, 3), "hz": self.hz,
                "source": None if self.source is None else f"{self.source.name} (device matched by {self.source.matched_by})",
                **({"error": self.error} if self.source is None and self.error else {})}

    class _Window:
        def __init__(self, tel: "GpuTelemetry") -> None:
            self.tel, self.t0, self.t1 = tel, 0.0, 0.0

        def __enter__(self) -> "GpuTelemetry._Window":
            self.t0 = time.perf_counter()
            return self

        def __exit__(self, *exc: Any) -> None:
            self.t1 = time.perf_counter()

        def summary(self) -> Dict[str, Any]:
            return self.tel.summary(self.t0, self.t1 or time.perf_counter())

    def window(self) -> "GpuTelemetry._Window":
        return GpuTelemetry._Window(self)


if __name__ == "__main__":
    import torch

    print("torch pci address of device 0:", pci_address_of(0))
    try:
        import amdsmi

        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        print("amdsmi handles:", len(hs))
        for h in hs:
            print("  bdf", amdsmi.amdsmi_get_gpu_device_bdf(h))
            for fn, args in (("amdsmi_get_power_info", ()), ("amdsmi_get_power_cap_info", ()),
                             ("amdsmi_get_clock_info", (amdsmi.AmdSmiClkType.GFX,))):
                try:
                    print("  ", fn, getattr(amdsmi, fn)(h, *args))
                except Exception as e:
                    print("  ", fn, "failed:", type(e).__name__, e)
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                print("   metrics:", {k: v for k, v in m.items() if any(s in k for s in ("gfxclk", "socket_power", "hotspot", "throttle", "activity"))})
            except Exception as e:
                print("   metrics failed:", type(e).__name__, e)
    except Exception as e:
        print("amdsmi unusable:", type(e).__name__, e)
    tel = GpuTelemetry(0).start()
    print("source:", None if tel.source is None else tel.source.name, "error:", tel.error)
    x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    with tel.window() as w:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 3.0:
            for _ in range(50):
                x @ x
            torch.cuda.synchronize()
    print("3 s of 8192^3 bf16 matmuls:", w.summary())
    time.sleep(1.0)
    print("idle sample:", tel.sample_now())
    tel.stop()
