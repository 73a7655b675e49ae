"""Run statistics -> `logger/<key>.log` text files (reference: ovo/entities/logger.py:9-106).

Same class name, methods and files on disk: one text file per statistic, one value per line (`str(value)`), `n_obj` kept in
memory only.  Weights & Biases is optional in the reference and absent here: `use_wandb=True` raises unless the package
imports.  Host code; the only GPU calls are the memory queries.
"""
from __future__ import annotations

import pprint
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch

STAT_KEYS = ("frame_id", "t_sam", "t_obj", "n_obj", "n_matches", "t_up", "t_seg", "t_clip", "avg_fps", "ram", "vram", "spf")
_GB = 1000 ** 3                      # the reference reports decimal gigabytes (logger.py:64-65)


class Logger:
    def __init__(self, output_path: str, pid: Optional[int] = None, use_wandb: bool = False) -> None:
        import psutil
        self.output_path = Path(output_path)
        for sub in ("logger", "logger/segment_vis"):
            (self.output_path / sub).mkdir(parents=True, exist_ok=True)
        self.stats: Dict[str, List[Any]] = {k: [] for k in STAT_KEYS}
        self.python_process = psutil.Process(pid)
        self.use_wandb = use_wandb
        self._wandb = None
        if use_wandb:
            import wandb                                     # not installed in the offline image: fail here, not mid-run
            self._wandb = wandb

    def _publish(self, payload: Dict[str, Any]) -> None:
        if self._wandb is not None:
            self._wandb.log(payload)

    def log_ovo_stats(self, stats: Dict[str, Any], print_output: bool = False) -> None:
        """Per-keyframe timings / counts from `OVO` (ovo.py:152-164, 352-364).  Unknown keys raise KeyError, as upstream."""
        for key, value in stats.items():
            self.stats[key].append(value)
        self._publish({f"Semantic/{k}": v for k, v in stats.items()})
        if self._wandb is not None and "n_obj" in stats:
            for i, n in enumerate(stats["n_obj"]):
                self._publish({"Semantic/Frame": stats["frame_id"], f"Semantic/n_obj_{i}": n})
        if print_output:
            pprint.pprint(stats, width=160, compact=True)

    def log_fps(self, avg_fps: float) -> None:
        self.stats["avg_fps"].append(avg_fps)
        self._publish({"Semantic/avg_fps": avg_fps})

    def log_spf(self, spf: float) -> None:
        self.stats["spf"].append(spf)

    def log_memory_usage(self, frame_id: int) -> None:
        vram = 0.0
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            vram = torch.cuda.memory_allocated("cuda") / _GB
        ram = self.python_process.memory_info().rss / _GB
        self.stats["vram"].append(vram)
        self.stats["ram"].append(ram)
        self._publish({"Semantic/Frame": frame_id, "Semantic/vram": vram, "Semantic/ram": ram})

    def log_max_memory_usage(self) -> None:
        peak = 0.0
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            peak = torch.cuda.max_memory_allocated("cuda") / _GB
        self.stats["max_vram"] = [peak]
        self.stats["max_ram"] = [float(np.asarray(self.stats["ram"]).max())]

    def write_stats(self) -> None:
        for key, values in self.stats.items():
            if key == "n_obj":
                continue
            (self.output_path / "logger" / f"{key}.log").write_text("\n".join(str(v) for v in values))

    def print_final_stats(self) -> None:
        skip = ("frame_id", "max_vram", "max_ram")
        out = {f"Avg {k}": (np.asarray(v).mean().round(3) if len(v) else float("nan"))      # never-logged statistic: nan, as upstream
               for k, v in self.stats.items() if k not in skip and k != "n_obj"}               # (n_obj holds ragged lists)
        if "max_ram" in self.stats:
            out["Max RAM"] = round(self.stats["max_ram"][0], 2)
            out["Max vRAM"] = round(self.stats["max_vram"][0], 2)
        print("Final statistics:")
        pprint.pprint(out, compact=True)
