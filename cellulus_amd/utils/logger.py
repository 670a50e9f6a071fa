"""`Logger` — keeps the reference's outputs (``<title>.csv`` with an index column,
``<title>.png``; cellulus/utils/logger.py:7-35) but appends instead of
rewriting the whole history every iteration, and draws the plot only every
``plot_every`` calls: at GPU step rates the reference's per-iteration
CSV + PNG rewrite would dominate the step.  The PNG is rendered by a
background thread from a snapshot of the data (matplotlib needs 0.1-0.2 s per figure: in
the training thread that is four steps at the benchmark configuration; the thread mostly
waits for the device with the GIL released, so the render overlaps)."""

import threading
from typing import Dict, List


class Logger:
    def __init__(self, keys: List[str], title: str, plot_every: int = 100):
        self.keys = keys
        self.title = title
        self.plot_every = max(1, int(plot_every))
        self.data: Dict[str, List[float]] = {k: [] for k in keys}
        self._written = 0
        self._plots = 0
        self._window = None
        self._plot_thread = None
        self._plot_lock = threading.Lock()
        print(f"Created logger with keys: {keys}")

    def add(self, key, value):
        assert key in self.data, "Key not in data"
        self.data[key].append(value)

    def write(self):
        """<title>.csv: header ',key1,key2', then 'row_index,value1,value2' (pandas to_csv layout)."""
        n = min(len(v) for v in self.data.values()) if self.data else 0
        path = self.title + ".csv"
        if self._written == 0 or self._written > n:
            with open(path, "w") as f:
                f.write("," + ",".join(self.data.keys()) + "\n")
            self._written = 0
        if n > self._written:
            with open(path, "a") as f:
                for i in range(self._written, n):
                    f.write(str(i) + "," + ",".join(repr(float(self.data[k][i])) for k in self.data) + "\n")
            self._written = n

    def plot(self, force: bool = False):
        self._plots += 1
        if not force and self._plots % self.plot_every != 1 and self.plot_every != 1:
            return
        worker = getattr(self, "_plot_thread", None)
        if worker is not None and worker.is_alive():
            if not force:
                return                      # a render is still running: the next due plot shows these points too
            worker.join()
        snapshot = {k: list(v) for k, v in self.data.items()}
        if force:
            self._render(snapshot)
            return
        self._plot_thread = threading.Thread(target=self._render, args=(snapshot,), name="clx-plot", daemon=True)
        self._plot_thread.start()

    def _render(self, data):
        import matplotlib

        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt

        with self._plot_lock:
            if self._window is None:
                self._window = plt.subplots()
            fig, ax = self._window
            ax.cla()
            for key, values in data.items():
                ax.plot(range(len(values)), values, marker=".")
            ax.set_xlabel("Iteration")
            ax.set_ylabel(self.title)
            fig.savefig(self.title + ".png")
