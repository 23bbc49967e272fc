"""Several batches in flight on one GPU: independent lanes, each with its own encoder handle, HIP stream and host thread.

The reference handles one request at a time, start to finish (`backend/app/main.py:177-232`: encode, then
`collection.query`, inside one route call). A batch's encode is a dependent chain of ~75 launches; each GEMM of the chain
ends with a partly filled last round of tiles and a store burst during which the matrix cores of the drained CUs idle, and
the search that follows is a handful of short, latency-bound kernels plus a host read-back. Nothing INSIDE one batch can
fill those holes (every kernel depends on the previous one), but another batch can: with two lanes the hardware dispatcher
places the other lane's workgroups on the CUs a kernel's tail leaves free. Measured on MI355X, ViT-B/32, 256 images per
batch, encode + top-10 against 100 k rows (`tools/two_batches_probe.py`): 3.21 -> 2.96 ms per batch (+8 % images/s) with two
lanes; a third lane gives nothing more. Splitting ONE batch over two streams loses instead (half-height GEMMs leave
half of the CUs without a tile: `tools/dual_stream_probe.py`), and so does overlapping only the search stage with the next
encode (the search kernels are too short to matter and mostly wait for a CU: `tools/staged_pipeline_probe.py`).

Results are bit-identical to running the batches one after another: same kernels, same inputs, only the stream differs.
The C-ABI contract this relies on (`include/mmiss.h`): handles are independent and re-entrant across handles; one handle
serialises its own calls. The index handle may be shared by all lanes (its calls serialise, the encodes do not).

Single process, single GPU. With several ranks keep collectives out of `work`: lanes run in no fixed order.
"""
from __future__ import annotations

import queue
import threading
from typing import Any, Callable, List, Optional, Sequence, Tuple

import torch


class BatchLanes:
    """`work(lane, item)` runs in lane `lane`'s thread with that lane's stream current; it typically calls
    `encoders[lane].encode_image(...)` and a search on the result, and may block (ctypes calls drop the GIL)."""

    def __init__(self, n_lanes: int, work: Callable[[int, Any], Any], device=0, queue_depth: int = 2):
        if n_lanes < 1:
            raise ValueError("n_lanes must be >= 1")
        self._device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self._work = work
        self._streams = [torch.cuda.Stream(device=self._device) for _ in range(n_lanes)]
        self._queues: List["queue.Queue[Optional[Tuple[int, Any]]]"] = [queue.Queue(maxsize=queue_depth) for _ in range(n_lanes)]
        self._results: dict = {}
        self._lock = threading.Lock()
        self._error: Optional[BaseException] = None
        self._submitted = 0
        self._collected = 0
        self._threads = [threading.Thread(target=self._run, args=(i,), name=f"mmiss-lane-{i}", daemon=True)
                         for i in range(n_lanes)]
        for t in self._threads:
            t.start()

    @property
    def n_lanes(self) -> int:
        return len(self._streams)

    def _run(self, lane: int) -> None:
        torch.cuda.set_device(self._device)
        q = self._queues[lane]
        with torch.cuda.stream(self._streams[lane]):
            while True:
                job = q.get()
                try:
                    if job is None:
                        return
                    seq, item = job
                    if self._error is None:
                        r = self._work(lane, item)
                        with self._lock:
                            self._results[seq] = r
                except BaseException as e:  # re-raised by drain(); the lanes keep consuming so that nothing blocks
                    with self._lock:
                        if self._error is None:
                            self._error = e
                finally:
                    q.task_done()

    def submit(self, item: Any = None) -> int:
        """Hand one batch to the next lane (round robin); blocks only while that lane's queue is full."""
        seq = self._submitted
        self._submitted += 1
        self._queues[seq % self.n_lanes].put((seq, item))
        return seq

    def drain(self) -> List[Any]:
        """Wait for every submitted batch; returns the results of the batches submitted since the last drain, in
        submission order."""
        for q in self._queues:
            q.join()
        for s in self._streams:
            s.synchronize()
        with self._lock:
            err, self._error = self._error, None
            res, self._results = self._results, {}
        first, self._collected = self._collected, self._submitted
        if err is not None:
            raise err
        return [res[i] for i in range(first, self._submitted)]

    def map(self, items: Sequence[Any]) -> List[Any]:
        for it in items:
            self.submit(it)
        return self.drain()

    def close(self) -> None:
        for q, t in zip(self._queues, self._threads):
            if t.is_alive():
                q.put(None)
        for t in self._threads:
            t.join()

    def __enter__(self) -> "BatchLanes":
        return self

    def __exit__(self, *exc) -> None:
        self.close()
