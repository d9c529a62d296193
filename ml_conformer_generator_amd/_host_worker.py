"""Worker process of `host_pool.HostPool`.  Started as a FRESH interpreter (`python _host_worker.py <task_fd> <result_fd>`),
run BY FILE PATH so that the package's `__init__` (torch, the HIP library) is never imported here: a worker holds the
standard library, whatever the task file imports (numpy + RDKit for `_rdkit_tasks.py`) and nothing else - no GPU context,
no fork of a process that has one.

`sys.path[0]` - this package's own directory, because the script is run by path - is REMOVED first: left in place, the
package's `config.py`, `distributed.py`, `schedule.py`, `weights.py` ... would shadow top-level modules of the same name
for everything a task imports (`import distributed` inside a caller's task file would pull in torch and this package).

Protocol (two dedicated pipes; stdout is pointed at stderr so a chatty task cannot corrupt a frame):
    parent -> worker   pickle frame (task_file, func_name, items, args)       EOF = shut down
    worker -> parent   8-byte little-endian length + pickle of (True, results) | (False, exception, traceback_text)
                       (length-prefixed so that the parent can read it under a deadline, `host_pool._Worker.call`)
An exception whose class lives in a task file loaded by path (module `_mcg_host_task_<n>`: the parent cannot import it)
travels as a RuntimeError naming the original type.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != _HERE]

import importlib.util  # noqa: E402
import pickle  # noqa: E402
import struct  # noqa: E402
import traceback  # noqa: E402

_MODULES = {}


def _load(path: str):
    mod = _MODULES.get(path)
    if mod is None:
        name = "_mcg_host_task_%d" % len(_MODULES)
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        _MODULES[path] = mod
    return mod


def main() -> int:
    rd = os.fdopen(int(sys.argv[1]), "rb")
    wr = os.fdopen(int(sys.argv[2]), "wb")
    sys.stdout = sys.stderr
    for path in sys.argv[3:]:            # preload (host_pool.prestart): import the task files now, while the parent is busy elsewhere
        try:
            _load(path)
        except BaseException:  # noqa: BLE001 - the task that needs the file reports the error properly
            _MODULES.pop(path, None)
    while True:
        try:
            msg = pickle.load(rd)
        except EOFError:
            return 0
        try:
            path, func, items, args = msg
            reply = (True, getattr(_load(path), func)(items, *args))
            data = pickle.dumps(reply, protocol=pickle.HIGHEST_PROTOCOL)
        except BaseException as e:  # noqa: BLE001 - reported to the parent, which re-raises it in the caller
            tb = traceback.format_exc()
            try:
                if (type(e).__module__ or "").startswith("_mcg_host_task_"):
                    raise TypeError("exception class defined in a by-path task module")      # the parent cannot import it
                data = pickle.dumps((False, e, tb), protocol=pickle.HIGHEST_PROTOCOL)
                pickle.loads(data)              # (an exception with a non-trivial __init__ pickles but does not rebuild)
            except Exception:  # noqa: BLE001 - an exception that does not make the trip
                data = pickle.dumps((False, RuntimeError(f"{type(e).__name__}: {e}"), tb), protocol=pickle.HIGHEST_PROTOCOL)
            if isinstance(e, (KeyboardInterrupt, SystemExit)):
                wr.write(struct.pack("<Q", len(data)) + data); wr.flush()
                return 1
        wr.write(struct.pack("<Q", len(data)) + data)
        wr.flush()


if __name__ == "__main__":
    sys.exit(main())
