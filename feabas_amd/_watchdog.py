"""A resident-set watchdog for runs on a shared GPU box: a thread polls this process's RSS (and that of its children, which a
test may spawn) and ends the process with os._exit when it passes a limit, so that a runaway host allocation dies as ONE
process with a message instead of as the machine (round 5 lost two boxes to a host OOM).  An address-space limit
(RLIMIT_AS / ulimit -v) is not an option: the HIP runtime reserves very large virtual ranges when it initialises."""
import os
import sys
import threading
import time

_started = None


def _rss_kb(pid):
    try:
        with open('/proc/%s/statm' % pid) as f:
            return int(f.read().split()[1]) * (os.sysconf('SC_PAGE_SIZE') // 1024)
    except (OSError, ValueError, IndexError):
        return 0


def _children(pid):
    out = []
    try:
        for t in os.listdir('/proc/%s/task' % pid):
            with open('/proc/%s/task/%s/children' % (pid, t)) as f:
                out += [int(c) for c in f.read().split()]
    except (OSError, ValueError):
        pass
    return out


def tree_rss_gb(pid=None):
    """RSS of `pid` (default: this process) plus all of its descendants, in GB."""
    pid = os.getpid() if pid is None else pid
    todo, kb, seen = [pid], 0, set()
    while todo:
        p = todo.pop()
        if p in seen:
            continue
        seen.add(p)
        kb += _rss_kb(p)
        todo += _children(p)
    return kb / 1048576.0


def start(limit_gb=None, period=0.25, exit_code=3):
    """Start the watchdog once per process; `limit_gb` defaults to $FEABAS_RSS_LIMIT_GB or 24.  Returns the limit in force
    (0 = disabled by FEABAS_RSS_LIMIT_GB=0)."""
    global _started
    if _started is not None:
        return _started
    if limit_gb is None:
        limit_gb = float(os.environ.get('FEABAS_RSS_LIMIT_GB', '24'))
    _started = limit_gb
    if limit_gb <= 0:
        return 0

    def run():
        while True:
            g = tree_rss_gb()
            if g > limit_gb:
                try:
                    sys.stderr.write('\n[feabas_amd watchdog] resident set %.1f GB > limit %.1f GB: ending the process '
                                     '(FEABAS_RSS_LIMIT_GB changes the limit)\n' % (g, limit_gb))
                    sys.stderr.flush()
                    import faulthandler
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                    sys.stderr.flush()
                finally:
                    os._exit(exit_code)
            time.sleep(period)

    threading.Thread(target=run, name='fb-rss-watchdog', daemon=True).start()
    return limit_gb


_BACKSTOP = r'''
import os, signal, sys, time
def _rss_kb(pid):
    try:
        with open('/proc/%%s/statm' %% pid) as f:
            return int(f.read().split()[1]) * (os.sysconf('SC_PAGE_SIZE') // 1024)
    except (OSError, ValueError, IndexError):
        return 0
def _children(pid):
    out = []
    try:
        for t in os.listdir('/proc/%%s/task' %% pid):
            with open('/proc/%%s/task/%%s/children' %% (pid, t)) as f:
                out += [int(c) for c in f.read().split()]
    except (OSError, ValueError):
        pass
    return out
def tree_rss_gb(pid):
    todo, kb, seen = [pid], 0, set()
    while todo:
        p = todo.pop()
        if p in seen:
            continue
        seen.add(p); kb += _rss_kb(p); todo += _children(p)
    return kb / 1048576.0
ppid, limit = %(pid)d, %(limit)r
while True:
    if os.getppid() != ppid:
        sys.exit(0)
    g = tree_rss_gb(ppid)
    if g > limit:
        sys.stderr.write('\n[feabas_amd watchdog/backstop] resident set %%.1f GB > %%.1f GB: killing %%d\n' %% (g, limit, ppid))
        sys.stderr.flush()
        kids = [c for c in _children(ppid) if c != os.getpid()]
        for p in kids + [ppid]:
            try:
                os.kill(p, signal.SIGKILL)
            except OSError:
                pass
        sys.exit(3)
    time.sleep(0.2)
'''


def start_backstop(limit_gb=None):
    """The same limit enforced from a CHILD process (a thread of this process cannot run while a C call holds the GIL): polls
    this process tree's RSS and SIGKILLs it above 1.15 x the limit.  Start it before anything touches the GPU (it is a plain
    child process, not an exec).  Ends by itself when this process does."""
    import subprocess
    if limit_gb is None:
        limit_gb = float(os.environ.get('FEABAS_RSS_LIMIT_GB', '24'))
    if limit_gb <= 0:
        return None
    code = _BACKSTOP % dict(pid=os.getpid(), limit=1.15 * limit_gb)       # (self-contained: the child imports nothing of the package)
    return subprocess.Popen([sys.executable, '-S', '-c', code], stdin=subprocess.DEVNULL, close_fds=True)
