"""Utterance-level data parallelism: streams are independent (no cross-stream term anywhere on the
path), so a batch is partitioned across ranks with NO data-path collective.  torch.distributed
(RCCL over xGMI on the GPU box, gloo in CPU tests) is used only for the timing barrier and the
8-byte throughput reduction."""
import os
import time


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(total, rank, world):
    """Contiguous, balanced partition of `total` streams: rank r owns [lo, hi)."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request total=%d rank=%d world=%d" % (total, rank, world))
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_seed(seed, rank):
    """Independent synthetic data per shard (seed + rank)."""
    return seed + rank


def barrier(dist, device=None):
    if dist is not None and dist.is_initialized():
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def reduce_throughput(dist, frames, seconds, device):
    """(total frames over all ranks, max seconds over ranks)."""
    import torch
    if dist is None or not dist.is_initialized():
        return int(frames), float(seconds)
    f = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    s = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(s, op=dist.ReduceOp.MAX)
    return int(round(f.item())), float(s.item())


def count_ranks(dist, device):
    """all-reduce(SUM) of 1: how many ranks actually took part in the report (the bench line's `ranks_seen`)."""
    import torch
    if dist is None or not dist.is_initialized():
        return 1
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def rank_identity(rank, local_rank, device, frames_per_s):
    """What one rank reports about itself in the bench line: host, the GPU it drove (uuid, else PCI bus id) and its own rate."""
    import socket
    import torch
    info = {"rank": int(rank), "local_rank": int(local_rank), "host": socket.gethostname(), "device": str(device),
            "device_id": None, "device_name": None, "mel_frames_per_s": float(frames_per_s)}
    if getattr(device, "type", "cpu") == "cuda":
        props = torch.cuda.get_device_properties(device)
        info["device_name"] = props.name
        ident = getattr(props, "uuid", None)
        if ident is None or not str(ident).strip("0-"):
            ident = "pci:%s:%s:%s" % (getattr(props, "pci_domain_id", "?"), getattr(props, "pci_bus_id", "?"),
                                      getattr(props, "pci_device_id", "?"))
        info["device_id"] = str(ident)
    else:
        info["device_id"] = "%s:pid%d" % (device, os.getpid())
    return info


def find_sclk_path(device):
    """sysfs file with the shader-clock table of `device` (pp_dpm_sclk of the DRM card with the device's PCI address), or
    None: not a GPU, or no such file.  Resolved once, OUTSIDE any timed region (imports, a device-property query, a glob)."""
    if getattr(device, "type", "cpu") != "cuda":
        return None
    try:
        import glob
        import re
        import torch
        props = torch.cuda.get_device_properties(device)
        want = "%04x:%02x:%02x." % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
        for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                slot = re.search(r"PCI_SLOT_NAME=(\S+)", open(os.path.join(card, "uevent")).read())
                if slot and slot.group(1).lower().startswith(want) and os.path.exists(os.path.join(card, "pp_dpm_sclk")):
                    return os.path.join(card, "pp_dpm_sclk")
            except Exception:
                continue
    except Exception:
        pass
    return None


def read_sclk_file(path):
    """The starred line of a pp_dpm_sclk file in MHz, or None (not readable as this user)."""
    import re
    try:
        for ln in open(path):
            m = re.match(r"\s*\d+:\s*(\d+)\s*[Mm][Hh]z\s*\*", ln)
            if m:
                return int(m.group(1))
    except Exception:
        pass
    return None


def read_sclk_mhz(device):
    """Current shader clock of `device` in MHz, or None.  Informational only (bench.py per_rank)."""
    path = find_sclk_path(device)
    return read_sclk_file(path) if path else None


class ClockSampler(object):
    """Samples the shader clock ONCE, `delay_s` after start(), from a helper thread: the timed loop of bench.py never
    executes the sysfs read itself (the read is an SMU query; done inline it sat inside the timed interval)."""

    def __init__(self, device, delay_s):
        import threading
        self.path, self.delay_s, self.value = find_sclk_path(device), max(0.0, float(delay_s)), None
        self._thread = threading.Thread(target=self._run, daemon=True) if self.path else None

    def _run(self):
        time.sleep(self.delay_s)
        self.value = read_sclk_file(self.path)

    def start(self):
        if self._thread is not None:
            self._thread.start()

    def result(self):
        if self._thread is not None:
            self._thread.join(2.0)
        return self.value


def gather_rank_info(dist, info):
    """[info of rank 0, ..., info of rank world-1] on every rank (all_gather_object after the timed region)."""
    if dist is None or not dist.is_initialized():
        return [info]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, info)
    return out
