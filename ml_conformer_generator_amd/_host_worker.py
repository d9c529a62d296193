"""Worker process of `host_pool.HostPool`.  Started as a FRESH interpreter (`python _host_worker.py <task_fd> <result_fd>`),
run BY FILE PATH so that the package's `__init__` (torch, the HIP library) is never imported here: a worker holds the
standard library, whatever the task file imports (numpy + RDKit for `_rdkit_tasks.py`) and nothing else - no GPU context,
no fork of a process that has one.

Protocol (pickle frames over two dedicated pipes; stdout is pointed at stderr so a chatty task cannot corrupt a frame):
    parent -> worker   (task_file, func_name, items, args)       EOF = shut down
    worker -> parent   (True, results) | (False, exception, traceback_text)
"""
import importlib.util
import os
import pickle
import sys
import traceback

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
                data = pickle.dumps((False, e, tb), protocol=pickle.HIGHEST_PROTOCOL)
            except Exception:  # noqa: BLE001 - an exception that does not pickle
                data = pickle.dumps((False, RuntimeError(f"{type(e).__name__}: {e}"), tb), protocol=pickle.HIGHEST_PROTOCOL)
            if isinstance(e, (KeyboardInterrupt, SystemExit)):
                wr.write(data); wr.flush()
                return 1
        wr.write(data)
        wr.flush()


if __name__ == "__main__":
    sys.exit(main())
