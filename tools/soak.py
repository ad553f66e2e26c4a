#!/usr/bin/env python3
"""Soak of the control loop: thousands of batches, device memory and host RSS at the start and at the end (a loop that allocates
per batch shows here), latency drift between the first and the last thousand batches.
    python tools/soak.py [--channels 512] [--batches 20000] [--dtype f32w]"""
import argparse, os, resource, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from riser_amd import synth
from riser_amd.replay import scripted_batches
from riser_amd.fake_client import FakeClient


class SoakClient(FakeClient):
    """drops what it has played and keeps no record of the calls: the soak measures the loop, not the script"""

    def get_read_batch(self):
        b = super().get_read_batch()
        self._batches[self._next - 1] = None
        return b

    def reject_reads(self, reads, unblock_duration):
        pass

    def finish_processing_reads(self, reads):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--batches", type=int, default=20000)
    ap.add_argument("--dtype", default="f32w")
    args = ap.parse_args()
    import logging, tempfile
    from riser_amd import Model, SignalProcessor, Kit
    from riser_amd.control import SequencerControl
    dev = torch.device("cuda", 0)
    models = [Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=args.dtype, device=dev)]
    proc = SignalProcessor(Kit.create_from_version("RNA004"), device=dev)
    period = 64                                   # the script repeats: read ids differ per turn, so a repeat is new traffic
    log = logging.getLogger("soak"); log.addHandler(logging.NullHandler())
    with tempfile.TemporaryDirectory() as d:
        client = SoakClient([], 1, args.channels)
        ctl = SequencerControl(client, models, proc, log, os.path.join(d, "out"))
        ctl.start()
        marks = []
        done = 0
        while done < args.batches:
            client.extend(scripted_batches(period, args.channels, first_batch=done))
            ctl.target("enrich", 1.0, 0.9)
            done += period
            if done in (period * 4, ) or done >= args.batches or done % (period * 50) == 0:
                torch.cuda.synchronize()
                recent = np.asarray(list(ctl.batch_latencies)[-256:]) * 1e3
                marks.append((done, torch.cuda.memory_allocated(dev) >> 20, torch.cuda.memory_reserved(dev) >> 20,
                              resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10, float(np.median(recent))))
        ctl.finish()
        lat = np.asarray(ctl.batch_latencies) * 1e3
    for m in marks:
        print("after %6d batches: device allocated %d MiB, reserved %d MiB, host max RSS %d MiB, p50 of the last 256 batches %.3f ms" % m)
    k = min(1000, lat.size // 3)
    print(f"latency p50 first {k}: {np.median(lat[8:k]):.3f} ms, last {k}: {np.median(lat[-k:]):.3f} ms, max overall {lat[8:].max():.3f} ms over {lat.size} batches")


if __name__ == "__main__":
    main()
