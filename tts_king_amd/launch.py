"""One process per GPU, started by the parent itself: `python bench.py --gpus N` (and `python train.py` with
`mi355x.gpus: N`) must not depend on being wrapped in torch.distributed.run.

The parent never initialises the GPU (on this pool a process that has may not start another program): it counts the
devices (`torch.cuda.device_count()` reads the driver's list without creating a context), starts N children of the same
command line with the torch.distributed.run environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT, rendezvous on 127.0.0.1), relays rank 0's standard output and exits with the first non-zero child status.

reference: the reference has no launcher (train.py:104 is a commented-out nn.DataParallel); this replaces the
`torchrun` wrapper the driver would otherwise have to supply.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    """A port nobody listens on right now.  (Bind-and-close leaves a window in which another process may take it; spawn_ranks
    retries the whole job once on another port when rank 0 reports EADDRINUSE before it has produced any output.)"""
    s = socket.socket()
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Terminated(Exception):
    pass


def _raise_terminated(signum, frame):
    raise _Terminated(signum)


def under_profiler(env=None):
    env = os.environ if env is None else env
    return "rocprof" in env.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in env)


def wants_spawn(n_gpus, env=None):
    """True when this process was asked for N > 1 GPUs and is not already one of N ranks."""
    env = os.environ if env is None else env
    return int(n_gpus) > 1 and "WORLD_SIZE" not in env and "RANK" not in env


def visible_devices():
    """Number of HIP devices without creating a context (safe before starting children)."""
    import torch
    try:
        return int(torch.cuda.device_count())
    except Exception:                   # no driver at all
        return 0


def threading_main():
    import threading
    return threading.current_thread() is threading.main_thread()


def _signal_group(p, sig):
    """Signal the rank's whole session (it was started with start_new_session: its pgid is its pid); the exact group we started,
    never a pattern."""
    try:
        os.killpg(p.pid, sig)
    except (ProcessLookupError, PermissionError):
        try:
            p.send_signal(sig)
        except ProcessLookupError:
            pass


def spawn_ranks(n, argv, env=None, timeout=None, n_devices=None, stdout=None):
    """Start `argv` n times (rank r gets RANK=LOCAL_RANK=r), wait for all, write rank 0's stdout to `stdout` (default
    sys.stdout) and return the exit status: 0 only if every rank returned 0.  Ranks other than 0 have their stdout sent to
    stderr (the bench contract is ONE JSON line from rank 0).  If a rank fails the others are terminated (a missing rank would
    leave them waiting in the rendezvous or in a collective).  The rendezvous port is found by bind-and-close (`free_port`), which
    leaves a window for another process to take it: when the job fails with rank 0 reporting EADDRINUSE ("Address already in use")
    on its standard error before it has written anything to standard output, the whole job is started ONCE more on another port."""
    n = int(n)
    have = visible_devices() if n_devices is None else int(n_devices)
    if have < n:
        sys.stderr.write("asked for %d GPUs but %d HIP device(s) are visible: not starting\n" % (n, have))
        return 3
    if under_profiler(env):
        sys.stderr.write("running under rocprofv3 (GPU already initialised in this process): rank processes cannot be started "
                         "from here; profile with `rocprofv3 ... -- python3 -m torch.distributed.run ...` instead\n")
        return 4
    t_end = None if timeout is None else time.time() + timeout
    rc, out0 = 1, b""
    for attempt in range(2):
        rc, out0, port_taken = _run_ranks_once(n, argv, env, t_end)
        if rc == 0 or not port_taken or out0.strip() or rc >= 128 or rc == 124:
            break
        if attempt == 0:
            sys.stderr.write("rendezvous port was taken between its selection and rank 0's bind (EADDRINUSE): starting the job again on another port\n")
    (stdout or sys.stdout.buffer).write(out0)
    (stdout or sys.stdout.buffer).flush()
    return rc


_PORT_TAKEN = (b"EADDRINUSE", b"Address already in use", b"address already in use")


def _relay_stderr(pipe, seen):
    """Copy rank 0's standard error through to ours, remembering whether it ever named a taken rendezvous port."""
    out = getattr(sys.stderr, "buffer", None)
    tail = b""
    for chunk in iter(lambda: pipe.read1(65536) if hasattr(pipe, "read1") else pipe.read(65536), b""):
        if any(m in tail + chunk for m in _PORT_TAKEN):
            seen.append(True)
        tail = chunk[-32:]
        try:
            if out is not None:
                out.write(chunk); out.flush()
            else:
                sys.stderr.write(chunk.decode(errors="replace")); sys.stderr.flush()
        except (OSError, ValueError):
            pass
    pipe.close()


def _run_ranks_once(n, argv, env, t_end):
    """One attempt of the job on a fresh port: (exit status, rank 0's stdout, did rank 0 report a taken port)."""
    import threading
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
                 "HSA_ENABLE_IPC_MODE_LEGACY": base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    procs = []
    # SIGTERM / SIGINT / SIGHUP of the parent (`timeout N python bench.py`, a scheduler's kill) must not orphan the ranks — they would
    # keep their GPUs and hang in the rendezvous or a collective: the handlers raise into the `finally` below, which ends every rank;
    # the ranks live in a session of their own so that the whole tree of each (a rank's own helpers) can be signalled as a group.
    old = {}
    if threading_main():
        for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            old[sig] = signal.signal(sig, _raise_terminated)
    rc, out0 = 0, b""
    port_taken, relay = [], None
    try:
        for r in range(n):
            e = dict(base)
            e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
            procs.append(subprocess.Popen(list(argv), env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                          stderr=subprocess.PIPE if r == 0 else None, start_new_session=True))
        relay = threading.Thread(target=_relay_stderr, args=(procs[0].stderr, port_taken), daemon=True)
        relay.start()
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                p = procs[r]
                if r == 0:
                    # drain rank 0's pipe while waiting so that a large record cannot block it (its stderr is drained by the relay)
                    chunk = _read_available(p.stdout, 0.2)
                    out0 += chunk
                    if chunk or p.poll() is None:
                        continue
                    out0 += p.stdout.read() or b""
                elif p.poll() is None:
                    continue
                pending.discard(r)
                if p.returncode != 0 and rc == 0:
                    rc = p.returncode
            if rc != 0:
                break
            if t_end is not None and time.time() > t_end:
                rc = 124
                break
            if pending and 0 not in pending:
                time.sleep(0.1)
    except _Terminated as t:
        rc = 128 + int(t.args[0])
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)
        for p in procs:
            if p.poll() is None:
                _signal_group(p, signal.SIGTERM)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                _signal_group(p, signal.SIGKILL)
                p.wait()
        if relay is not None:
            relay.join(timeout=5)
    return rc, out0, bool(port_taken)


def _read_available(pipe, wait_s):
    """Up to 64 KiB of what `pipe` holds, waiting at most `wait_s` for the first byte; b"" when nothing came (or at end of file)."""
    import select
    try:
        ready, _, _ = select.select([pipe], [], [], wait_s)
    except (OSError, ValueError):
        return b""
    if not ready:
        return b""
    return os.read(pipe.fileno(), 65536)
