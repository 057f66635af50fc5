#!/usr/bin/env python3
"""What does the HIP runtime answer to an event QUERY while a stream captures?  (The question behind the fence that
GraphTrainer._capture used to keep: a process group's watchdog thread polls the end events of earlier EAGER collectives
while the training step -- with its collectives -- is being captured.)

Cases, each a plain API return code caught as a Python exception (nothing here can fault the device):
  A  event recorded eagerly on stream s, s then captures, queried from ANOTHER thread during the capture
  B  the same, queried from the CAPTURING thread
  C  event recorded eagerly on a side stream that JOINS the capture (waits for an event of the capturing stream), queried
     from another thread during the capture   <- RCCL's internal stream under a captured collective
  D  an event recorded INSIDE the capture, queried from another thread (the call PyTorch's process group never makes: it does
     not hand captured work to its watchdog)
python3 tools/event_query_probe.py [thread_local|global|relaxed] [cases, e.g. C or ACDB]  ->  one JSON line (one capture mode per process)
"""
import json
import sys
import threading

import torch


def query_in_thread(ev):
    out = {}

    def run():
        try:
            out['r'] = bool(ev.query())
        except Exception as e:                                  # noqa: BLE001
            out['r'] = 'raised: ' + repr(e)[:200]
    t = threading.Thread(target=run)
    t.start(); t.join()
    return out['r']


def main():
    dev = torch.device('cuda', 0)
    res = {}
    x = torch.zeros(1024, device=dev)
    which = sys.argv[2] if len(sys.argv) > 2 else 'ACDB'      # the cases to run (a refused query may invalidate the capture: one process per set)
    for mode in (sys.argv[1:2] or ['thread_local']):
        s, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            x.add_(1.0)
            ev_s = torch.cuda.Event(); ev_s.record(s)
        with torch.cuda.stream(side):
            x.add_(1.0)
            ev_side = torch.cuda.Event(); ev_side.record(side)
        torch.cuda.synchronize()
        r = {}
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
                x.add_(1.0)
                if 'A' in which:
                    r['A_other_thread_eager_event_of_capturing_stream'] = query_in_thread(ev_s)
                fork = torch.cuda.Event(); fork.record(s)
                side.wait_event(fork)                           # the side stream joins the capture
                with torch.cuda.stream(side):
                    x.add_(1.0)
                    inside = torch.cuda.Event(); inside.record(side)
                if 'C' in which:
                    r['C_other_thread_eager_event_of_joined_stream'] = query_in_thread(ev_side)
                if 'D' in which:
                    r['D_other_thread_event_recorded_inside_capture'] = query_in_thread(inside)
                s.wait_event(inside)                            # join back
                if 'B' in which:                                # last: a refused call on the capturing thread invalidates the capture
                    try:
                        r['B_capturing_thread_eager_event_of_capturing_stream'] = bool(ev_s.query())
                    except Exception as e:                      # noqa: BLE001
                        r['B_capturing_thread_eager_event_of_capturing_stream'] = 'raised: ' + repr(e)[:200]
            r['capture'] = 'ok'
            g.replay(); torch.cuda.synchronize()
            r['replay'] = 'ok'
        except Exception as e:                                  # noqa: BLE001
            r['capture'] = 'raised: ' + repr(e)[:300]
        res[mode] = r
        torch.cuda.synchronize()
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
