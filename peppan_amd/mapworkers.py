"""Worker processes for the genes -> genomes mapping: the reference's `pool.imap_unordered(iter_map_bsn, ...)` over its forked
workers (PEPPAN.py:907-923), for a path whose search runs on the GPU.

One genome costs about 1.5 ms of GPU time and ten times that of host bookkeeping (filters, overlaps, `build_groups`): ONE process keeps
an MI355X busy for a tenth of the time.  `MapWorkers(n)` starts n processes that each own a HIP context (stream, work space) on the same
device; rounds of genomes are dealt to whichever worker is free, every worker runs the batched search and `build_groups` for its round,
and the caller gets the per-genome columns (`GenomeGroups`) back IN JOB ORDER - what the stores need (PEPPAN.py:923, the in-order
variant).  The stores do not depend on the number of workers: a genome's groups do not depend on which other genomes shared its search
(tests/test_gpu_parity.py::test_get_map_bsn_batched_equals_per_genome_workers, ::test_map_workers_write_the_same_stores).

The workers are started as `python -m peppan_amd.mapworkers <socket>` (not forked: the parent may hold a HIP context; not through
multiprocessing's spawn either, which would import the caller's main module a second time) and talk over a unix socket with
multiprocessing.connection's pickled messages (form 'members': the BULK - the genomes' sequences out, gene-table rows and finished members
back - goes through files in a memory-backed scratch directory, one write and one read each, instead of being pickled through the
sockets by the keeping process's threads):
    parent -> worker   ('setup', {...})        ('round', k, jobs[, file])                  ('emit', k, first group id, file) | ('drop', k)     ('stop',)
    worker -> parent   ('ready', pid)          ('done', k, [GenomeGroups or StoreBlock])   ('error', k, traceback text)
                                               ('counts', k, groups per genome)  - form 'members' only, answered by 'emit' / 'drop' -
                                               ('done', k, what the stores take from the round: mapbsn.round_members)

Form 'members' moves the stores' work to the workers as well.  The .mat / .seq stores hold 1 000 consecutive groups per member, groups
numbered through all genomes - a round's first id is known once every round in front of it has been mapped.  So a worker reports its
round's group counts, is told the round's first id as soon as the rounds in front have reported theirs, and then makes every member that
lies inside its round completely (pickle stream, deflate, CRC: mapbsn.round_members); only the groups in front of its first and behind its
last member boundary travel as columns.  The keeping process appends finished payloads: 0.5 ms per genome instead of 7.  While a worker
makes the members of one round the next round is in its queue already, and a worker naps while it waits for the GPU it shares
(PEPPAN_HIP_SPIN_US=0: pep_event_wait).  Measured: DESIGN.md section 5, tools/map_pool_rate.py.
"""
import os
import pickle
import subprocess
import sys
import tempfile
import threading
import time
import traceback
from multiprocessing.connection import Client, Listener

__all__ = ['MapWorkers']


def _this_process_uses_the_gpu():
    """has this process made a HIP context (the drop-in's own, or torch's)?  Its hardware queues count against the GPU's slots like the workers'"""
    try:
        ub = sys.modules.get('peppan_amd.uberBlast')
        if ub is not None and any(k[0] == os.getpid() for k in getattr(ub, '_CTX', {})):
            return True
        torch = sys.modules.get('torch')
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return True


def _scratch_root():
    """where the bulk of the traffic between the processes lives for a moment: memory-backed if the machine offers it"""
    shm = '/dev/shm'
    try:
        if os.path.isdir(shm) and os.access(shm, os.W_OK | os.X_OK):
            st = os.statvfs(shm)
            if st.f_bavail * st.f_frsize >= 4 << 30:          # (a round of 16 genomes is ~70 MB there and back, two rounds per worker: a container's default 64 MB will not do)
                return shm
    except OSError:
        pass
    return None


def _jobs_to_file(path, jobs):
    """the sequences of a round's genomes back to back in `path`; returns what is left of the jobs (ids, lengths).  A sequence that is
    not text stays in the message"""
    meta = []
    with open(path, 'wb', buffering=0) as f:
        for id, taxon, seq in jobs:
            contigs = []
            for name, s in seq:
                if isinstance(s, str):
                    s = s.encode('ascii')
                if isinstance(s, (bytes, bytearray)):
                    f.write(s)
                    contigs.append((name, len(s), None))
                else:
                    contigs.append((name, -1, s))
            meta.append((id, taxon, contigs))
    return meta


def _jobs_from_file(path, meta):
    with open(path, 'rb', buffering=0) as f:
        data = f.read()
    os.unlink(path)
    jobs, at = [], 0
    for id, taxon, contigs in meta:
        seq = []
        for name, n, other in contigs:
            if n < 0:
                seq.append([name, other])
            else:
                seq.append([name, data[at:at + n]])                   # bytes: the search and K12 take them as they are (no decode, no second encode)
                at += n
        jobs.append((id, taxon, seq))
    return jobs


def _members_to_file(path, P):
    """the bulk of a round's result - gene-table rows, finished members - into `path`; the message keeps offsets"""
    with open(path, 'wb', buffering=0) as f:
        tab = P['table']
        f.write(memoryview(tab).cast('B') if tab.size else b'')
        at = tab.nbytes
        P['table'] = ('file', 0, int(tab.shape[0]), int(tab.shape[1]))
        for kind in ('mat', 'seq'):
            if P[kind] is None:
                continue
            refs = []
            for payload, crc, size, method in P[kind]['members']:
                f.write(payload)
                refs.append((at, len(payload), crc, size, method))
                at += len(payload)
            P[kind]['members'] = refs
    P['blob'] = path
    return P


def _members_from_file(P):
    import numpy as np
    path = P.pop('blob')
    with open(path, 'rb', buffering=0) as f:
        data = f.read()
    os.unlink(path)
    view = memoryview(data)
    _, at, rows, cols = P['table']
    P['table'] = np.frombuffer(data, dtype=np.int64, count=rows * cols, offset=at).reshape(rows, cols).copy()      # (the keeper adds to it in place)
    for kind in ('mat', 'seq'):
        if P[kind] is not None:
            P[kind]['members'] = [(view[at:at + n], crc, size, method) for at, n, crc, size, method in P[kind]['members']]
    return P


def _thread_cpu(names={}):
    """[(name, user + system CPU seconds)] of every thread of this process, the library's and the runtime's included (/proc/self/task), largest first"""
    tick, out = float(os.sysconf('SC_CLK_TCK')), []
    for tid in os.listdir('/proc/self/task'):
        try:
            stat = open('/proc/self/task/%s/stat' % tid).read()
        except OSError:
            continue
        comm, rest = stat[stat.index('(') + 1:stat.rindex(')')], stat[stat.rindex(')') + 2:].split()
        out.append((names.get(int(tid), comm) + ':' + tid, (int(rest[11]) + int(rest[12])) / tick))
    return sorted(out, key=lambda kv: -kv[1])


def _serve(address, authkey):
    """The worker's life: set-ups and rounds until 'stop' or until the parent goes away.

    Three threads.  The one that reads the socket hands 'round' messages to the SEARCH thread (batched search of both tools, K7, the filters and K11
    per genome) and 'emit' / 'drop' answers to the GROUPS thread (build_groups with K12 in a context of its own, the round's counts, its members, the
    'done' message).  A worker shares the GPU with seven others, and at the rate the pool maps the GPU is half busy: a search waits about as long for
    its turn as it computes, and those waits are where the round in front gets its groups and members made - the search thread sleeps inside the
    library (no interpreter lock held) while the groups thread works.  Rounds leave the worker in the order they came."""
    import queue
    conn = Client(address, family='AF_UNIX', authkey=authkey)
    conn.send(('hello', os.getpid()))                               # (which of the processes the pool started is behind this connection)
    sys.setswitchinterval(1e-3)                                     # (two threads that hold the interpreter lock for a few hundred microseconds at a time)
    state = {}
    clock, spent = time.perf_counter, dict(recv=0., search=0., groups=0., wait_first=0., members=0., send=0., genomes=0)
    todo, searched, replies = queue.Queue(), queue.Queue(maxsize=1), queue.Queue()
    send_lock = threading.Lock()
    profiles = []
    tids = {threading.get_native_id(): 'socket'}

    def say(msg):
        with send_lock:
            conn.send(msg)

    def profiled(f):
        if not os.environ.get('PEPPAN_WORKERS_PROFILE'):            # cProfile over this worker's rounds (a profile per thread), top functions on stderr at the end
            return f

        def g():
            import cProfile
            pr = cProfile.Profile()
            profiles.append(pr)
            pr.enable()
            try:
                f()
            finally:
                pr.disable()
        return g

    def searcher():
        while True:
            msg = todo.get()
            if msg is None:
                spent['search_thread_cpu'] = time.thread_time()
                searched.put(None)
                return
            k, jobs = msg[1], msg[2]
            t0 = clock()
            try:
                from . import mapbsn
                if len(msg) > 3 and msg[3] is not None:
                    jobs = _jobs_from_file(msg[3], jobs)
                search = state['search'] or (lambda *a: mapbsn._gpu_search(*a, genomes_per_batch=state['per_batch']))
                found = list(search(state['prefix'], state['clust'], jobs, state['params']))
                if len(found) != len(jobs):
                    raise RuntimeError('the search returned %d tables for %d genomes' % (len(found), len(jobs)))
                item = (k, jobs, found, None)
            except BaseException:
                item = (k, jobs, None, traceback.format_exc())
            spent['search'] += clock() - t0
            searched.put(item)

    def answer_for(k):
        """the parent's 'emit' / 'drop' for round k (answers for rounds that failed here are passed over); None: the worker is stopping"""
        while True:
            reply = replies.get()
            if reply is None or reply[1] == k:
                return reply

    def grouper():
        while True:
            item = searched.get()
            if item is None:
                spent['groups_thread_cpu'] = time.thread_time()
                return
            k, jobs, found, failed = item
            try:
                if failed is not None:
                    say(('error', k, failed))
                    continue
                from . import mapbsn
                t0 = clock()
                groups = mapbsn.build_groups_round([(blastab, overlap, seq) for (id, taxon, seq), (blastab, overlap) in zip(jobs, found)],
                                                   state['ortho'], state['old'], state['params'], state['ctx'])           # (ONE K12 call for the round)
                out = [G if state['form'] == 'groups' else mapbsn.StoreBlock(G) for G in groups]
                del groups
                del found, item
                spent['groups'] += clock() - t0
                spent['genomes'] += len(out)
                if state['form'] == 'members':
                    say(('counts', k, [B.n for B in out]))
                    t0 = clock()
                    reply = answer_for(k)
                    t1 = clock()
                    if reply is None:
                        return
                    out = mapbsn.round_members(out, [job[1] for job in jobs], reply[2], state['save_seq']) if reply[0] == 'emit' else None
                    if out is not None and len(reply) > 3 and reply[3] is not None:
                        out = _members_to_file(reply[3], out)
                    spent['wait_first'] += t1 - t0
                    spent['members'] += clock() - t1
                t0 = clock()
                say(('done', k, out))
                spent['send'] += clock() - t0
            except (EOFError, OSError):
                os._exit(1)                             # (the parent is gone)
            except BaseException:
                try:
                    say(('error', k, traceback.format_exc()))
                except Exception:
                    os._exit(1)

    threads = [threading.Thread(target=profiled(searcher), daemon=True), threading.Thread(target=profiled(grouper), daemon=True)]
    for t in threads:
        t.start()
    while True:
        try:
            t0 = clock()
            msg = conn.recv()
            spent['recv'] += clock() - t0
        except EOFError:
            os._exit(0)                                 # (the parent went away: nothing to finish, and the other threads may be inside the library)
        if msg[0] == 'stop':
            todo.put(None)
            replies.put(None)
            for t in threads:
                t.join(30.)
            if os.environ.get('PEPPAN_WORKERS_TIMING'):             # seconds this worker spent where (a line per worker on stderr)
                tm = os.times()
                spent['process_cpu_user'], spent['process_cpu_system'] = tm.user, tm.system
                sys.stderr.write('mapping worker %d: %s; threads still alive: %s\n' % (os.getpid(), ' '.join('%s %.2f' % kv for kv in spent.items()),
                                                                                        ' '.join('%s %.2f' % kv for kv in _thread_cpu(tids) if kv[1] >= 0.05)))
            if profiles:
                import pstats
                st = pstats.Stats(profiles[0], stream=sys.stderr)
                for pr in profiles[1:]:
                    st.add(pr)
                st.sort_stats('tottime').print_stats(30)
            return
        if msg[0] == 'setup':                           # (between two calls of the pool: both threads are idle)
            try:
                from . import mapbsn
                while not replies.empty():              # (round numbers start again with every call: nothing of an earlier one may be taken for an answer)
                    replies.get_nowait()
                a = msg[1]
                old = a['old_prediction']
                if state.get('old_is_mine'):
                    state['old'].close()
                ctx = a['ctx_class']() if a['ctx_class'] is not None else state.get('own_ctx')
                if ctx is None:
                    # the groups thread's K12 in a context of its own: the shared one (uberBlast.get_context) belongs to the search thread
                    from . import _native
                    ctx = state['own_ctx'] = _native.Context(int(os.environ.get('PEPPAN_HIP_DEVICE', os.environ.get('LOCAL_RANK', '0'))))
                state.update(a, ortho=mapbsn.OrthoRelation(a['orthoGroup']), old=mapbsn.MapBsn(old) if isinstance(old, str) else old,
                             old_is_mine=isinstance(old, str), ctx=ctx)
                say(('ready', os.getpid()))
            except BaseException:
                say(('error', -1, traceback.format_exc()))
        elif msg[0] == 'round':
            todo.put(msg)
        elif msg[0] in ('emit', 'drop'):
            replies.put(msg)


class _Hung(Exception):
    """a worker did not answer within its round's deadline"""


class MapWorkers(object):
    """n worker processes on one device.

        with MapWorkers(8) as pool:
            get_map_bsn(..., workers=pool)          # or workers=8: a pool for the length of that call

    `round_deadline` = (seconds, seconds per genome of the round): how long ONE answer of a worker may take.  A worker that DIES is noticed
    through its socket; one that HANGS (a GPU wait that never returns, a dead-locked library) would leave the call waiting for ever: it is
    killed, a FRESH child process is started in its place (never a restart of the process that held the GPU) and takes the round once
    more; a round that exceeds the deadline twice fails the call.  None: no deadline.  Default 120 s + 2 s per genome - a round of 16
    genomes takes a third of a second, a worker's first one a few seconds - or PEPPAN_WORKER_DEADLINE="seconds,seconds per genome".
    """

    def __init__(self, n, device=None, round_deadline='default'):
        self.n = int(n)
        if self.n < 1:
            raise ValueError('MapWorkers: at least one worker')
        if round_deadline == 'default':
            txt = os.environ.get('PEPPAN_WORKER_DEADLINE', '120,2')
            round_deadline = None if txt.strip().lower() in ('', 'none', '0') else tuple(float(x) for x in (txt.split(',') + ['0'])[:2])
        self.round_deadline = round_deadline
        self.replaced = 0                                   # workers killed for exceeding the deadline and replaced, over the pool's life
        self._dir = tempfile.mkdtemp(prefix='pep_workers_')
        self._bulk = tempfile.mkdtemp(prefix='pep_workers_', dir=_scratch_root())      # sequences out, members back: files, not messages
        self._address = address = os.path.join(self._dir, 's')
        authkey = os.urandom(16)
        self._listener = Listener(address, family='AF_UNIX', authkey=authkey)
        env = dict(os.environ, PEPPAN_WORKER_KEY=authkey.hex(), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                   PEPPAN_HOST_THREADS='1')             # (the workers ARE the parallelism: their host chain stays on the thread that runs it)
        env.setdefault('PEPPAN_HIP_SPIN_US', '0')           # the workers nap while they wait for the GPU they share: spinning would eat the CPU time the others need
        # few hardware queues per worker process: with the runtime's default (up to four per process) eight workers of three streams each - search, nucleotide
        # tool, the groups thread's K12 - ask for more queues than the GPU has slots for, and the scheduler spends its time swapping them: the driver's busy counter
        # reads 100 % at 330 genomes/s where it reads 44 % at 360 with one or two queues each (profiles/r05_pool_experiments.txt) ...
        # ... and two are what a worker needs: its two tools run one after the other and share ONE context (PEPPAN_ONE_CONTEXT: the second context of
        # uberBlast.get_nucl_context is for a process that runs them side by side), the groups thread's K12 has the other queue - on one queue a genome's K12
        # waited 4 ms behind the kernels of the search thread's next round (2 000 genomes: 452 -> 496 genomes/s)
        # - IF this process holds no queues of its own: a parent that has searched on the GPU itself (PEPPAN's has: get_similar_pairs runs before the mapping)
        # brings up to four, and 4 + 8 x 2 are over the limit again (the bench's pool leg: 383 genomes/s with the counter at 100 % against 423 with one queue each)
        env.setdefault('GPU_MAX_HW_QUEUES', '1' if _this_process_uses_the_gpu() else '2')
        env.setdefault('PEPPAN_ONE_CONTEXT', '1')
        if device is not None:
            env['PEPPAN_HIP_DEVICE'] = str(int(device))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env['PYTHONPATH'] = os.pathsep.join([root] + [p for p in sys.path if p] + [env.get('PYTHONPATH', '')])     # a search function of the caller's must be importable
        if os.environ.get('PEPPAN_POOL_GATE'):               # (experiment: at most that many workers inside their batched searches at once)
            env['PEPPAN_GPU_GATE'] = '%s,%d' % (self._dir, int(os.environ['PEPPAN_POOL_GATE']))
        self._env = env
        self._procs, self._conns, started = [], [], {}
        self._setup_msg, self._spawn_lock = None, threading.Lock()
        self.spent = dict(sequences_to_files=0., members_from_files=0.)        # seconds of this process's feeder threads (all of them together)
        try:
            try:
                self._listener._listener._socket.settimeout(120.)      # (a worker that dies before it connects must not leave accept() waiting for ever)
            except AttributeError:
                pass
            for _ in range(self.n):
                p = self._spawn()
                started[p.pid] = p
            for _ in range(self.n):
                conn = self._listener.accept()
                self._conns.append(conn)
                self._procs.append(started.pop(self._hello(conn)))        # connections arrive in any order: _procs[i] is the process behind _conns[i]
        except BaseException:
            self._procs += list(started.values())
            self.close()
            raise

    def _spawn(self):
        return subprocess.Popen([sys.executable, '-m', 'peppan_amd.mapworkers', self._address], env=self._env, stdin=subprocess.DEVNULL,
                                stdout=sys.stderr.fileno() if hasattr(sys.stderr, 'fileno') and self._has_fd(sys.stderr) else subprocess.DEVNULL)

    @staticmethod
    def _hello(conn):
        if not conn.poll(120.):
            raise RuntimeError('MapWorkers: a worker connected and did not say who it is')
        return conn.recv()[1]

    @staticmethod
    def _has_fd(stream):
        try:
            stream.fileno()
            return True
        except Exception:
            return False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        for c in self._conns:
            try:
                c.send(('stop',))
                c.close()
            except Exception:
                pass
        for p in self._procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
        self._conns, self._procs = [], []
        if self._listener is not None:
            self._listener.close()
            self._listener = None
        import shutil
        shutil.rmtree(self._bulk, ignore_errors=True)
        try:
            os.rmdir(self._dir)
        except OSError:
            pass

    def setup(self, prefix, clust, orthoGroup, old_prediction, params, search=None, per_batch=32, ctx_class=None, form='groups', save_seq=True):
        """what every round of one get_map_bsn call shares.  `search` must be None (the batched GPU search) or a function the workers can
        import (a module-level function: it is sent by reference); `ctx_class`: None (every worker uses its own HIP context) or an
        importable class whose instances stand in for one (the CPU tests' oracle-backed context); `form`: 'groups' - the workers return
        GenomeGroups -, 'stores' - they return what the stores take from a genome (mapbsn.StoreBlock), made on their side - or 'members'
        (module docstring; `save_seq`: whether there is a .seq store)"""
        if search is not None:
            try:
                pickle.dumps(search)
            except Exception:
                raise ValueError('MapWorkers: search must be None or a module-level function (it is sent to the workers by reference)')
        if not isinstance(old_prediction, str):
            old_prediction = getattr(old_prediction, 'fname', old_prediction)        # an open MapBsn is reopened by every worker
        msg = ('setup', dict(prefix=prefix, clust=clust, orthoGroup=orthoGroup, old_prediction=old_prediction, params=dict(params), search=search, per_batch=int(per_batch), ctx_class=ctx_class, form=form, save_seq=bool(save_seq)))
        self.form, self._setup_msg = form, msg
        for c in self._conns:
            c.send(msg)
        for c in self._conns:
            # the answer to THIS message: a worker that is still sending what an abandoned round left behind (a reused pool whose last call
            # failed half-way) is read past; a worker that says nothing within the deadline is an error, not a wait for ever
            deadline = time.monotonic() + 300.
            while True:
                if not c.poll(max(0., deadline - time.monotonic())):
                    raise RuntimeError('MapWorkers: a worker did not answer the set-up message within 300 s')
                r = c.recv()
                if r[0] in ('ready', 'error'):
                    break
            if r[0] != 'ready':
                raise RuntimeError('MapWorkers: a worker failed to set up:\n%s' % (r[2] if isinstance(r[2], str) else repr(r[2]),))

    def _replace(self, i):
        """worker i is given up: killed, and a fresh child process - started by this process like the first ones, set up like them - takes its place"""
        with self._spawn_lock:                          # (one at a time: the process started here is the one accepted here)
            old = self._procs[i]
            try:
                old.kill()
                old.wait(timeout=20)
            except Exception:
                pass
            try:
                self._conns[i].close()
            except Exception:
                pass
            self._procs[i] = self._spawn()
            conn = self._conns[i] = self._listener.accept()
            if self._hello(conn) != self._procs[i].pid:
                raise RuntimeError('MapWorkers: an unknown process connected in place of the replacement worker')
            self.replaced += 1
        conn.send(self._setup_msg)
        if not conn.poll(120.):
            raise RuntimeError('the replacement of a hung mapping worker did not come up')
        r = conn.recv()
        if r[0] != 'ready':
            raise RuntimeError('the replacement of a hung mapping worker failed to set up:\n' + r[2])
        return conn

    def rounds(self, jobs, per_round, first=0):
        """In job order: (job, GenomeGroups or StoreBlock) for every job - forms 'groups' / 'stores' - or (the round's jobs, what the stores
        take from the round) for every round - form 'members', `first` being the id of the first group of the first round.  Rounds of
        `per_round` jobs go to whichever worker is free; at most two rounds per worker are in flight or waiting to be taken.
        Whatever goes wrong in the threads that talk to the workers - a worker gone or hung, a sequence that cannot be written, a job that
        cannot be pickled - reaches the caller as RuntimeError for the first round concerned: no round is left waiting."""
        per_round = max(1, int(per_round))
        n_rounds = -(-len(jobs) // per_round)
        members = getattr(self, 'form', 'groups') == 'members'
        cond = threading.Condition()
        slots = threading.Semaphore(2 * self.n)
        state = dict(next=0, stop=False)
        results, counts, first_of = {}, {}, {0: int(first)}
        again = set()                                   # rounds that have been given to a replacement worker already

        def take_number():
            """the number of the next round (None when there is none, or the call is being given up)"""
            slots.acquire()
            with cond:
                k = state['next']
                if k >= n_rounds or state['stop']:
                    slots.release()
                    return None
                state['next'] = k + 1
            return k

        def send_round(i, k):
            mine = jobs[k * per_round:(k + 1) * per_round]
            if members:
                j_path = os.path.join(self._bulk, 'j%d' % k)
                t0 = time.perf_counter()
                meta = _jobs_to_file(j_path, mine)
                self.spent['sequences_to_files'] += time.perf_counter() - t0
                self._conns[i].send(('round', k, meta, j_path))
            else:
                self._conns[i].send(('round', k, mine))

        def answer(i, k):
            """the worker's next message about round k - within the round's deadline"""
            conn = self._conns[i]
            if self.round_deadline is not None:
                limit = self.round_deadline[0] + self.round_deadline[1] * len(jobs[k * per_round:(k + 1) * per_round])
                if not conn.poll(limit):
                    raise _Hung('no answer about round %d within %.0f s' % (k, limit))
            return conn.recv()

        def one_round(i, k, held, resend):
            """round k (in worker i's queue already) to its end -> the message for the caller.  Form 'members': the next round is handed
            out (held[1]; `resend`: the round that was held ahead when the worker had to be replaced) while the worker makes k's members"""
            if members and held[1] is None:                 # the round behind k goes into the worker's queue at once: its search runs while k's groups and members are made
                ahead = resend if resend is not None else take_number()
                if ahead is not None:
                    try:
                        send_round(i, ahead)
                        held[1] = ahead
                    except (EOFError, OSError):
                        held[1] = ahead
                        raise
                    except BaseException:                   # a round that cannot be handed out (its sequences, its pickle) fails the call - and round k, which IS
                        with cond:                          # in the worker, is still seen to its end: the pool stays in step
                            results[ahead] = ('error', ahead, traceback.format_exc())
                            state['stop'] = True
                            cond.notify_all()
            msg = answer(i, k)
            if msg[0] == 'counts':
                with cond:
                    counts[k] = sum(msg[2])
                    j = k
                    while j in counts and j in first_of:            # (every round whose predecessors have all reported now knows its first id)
                        first_of[j + 1] = first_of[j] + counts[j]
                        j += 1
                    cond.notify_all()
                    while k not in first_of and not state['stop']:
                        cond.wait()
                    go = ('emit', k, first_of[k], os.path.join(self._bulk, 'r%d' % k)) if k in first_of else ('drop', k)
                self._conns[i].send(go)
                msg = answer(i, k)
                if go[0] == 'drop':
                    msg = ('dropped', k, None)
                elif msg[0] == 'done' and msg[2] is not None and 'blob' in msg[2]:
                    t0 = time.perf_counter()
                    msg = ('done', k, _members_from_file(msg[2]))
                    self.spent['members_from_files'] += time.perf_counter() - t0
            return msg

        def feeder(i):
            # form 'members': while the worker makes the members of round k, the next round is in its queue already (its sequences written, the
            # message sent) - the worker never waits for this thread between two rounds
            held = [None, None]                     # the round this thread is waiting for, the round handed out ahead of its end
            resend = None
            try:
                held[0] = take_number()
                if held[0] is not None:
                    send_round(i, held[0])
                while held[0] is not None:
                    k = held[0]
                    try:
                        msg = one_round(i, k, held, resend)
                    except _Hung as e:
                        mine = [r for r in held if r is not None]
                        if again.intersection(mine):
                            try:
                                self._replace(i)            # (the pool stays usable: the process that hangs is not left behind in it)
                            except Exception:
                                pass
                            raise RuntimeError('a mapping worker hung twice: %s' % (e,))
                        again.update(mine)
                        sys.stderr.write('MapWorkers: %s - the worker (pid %d) is killed, a fresh process takes the round once more\n' % (e, self._procs[i].pid))
                        self._replace(i)
                        resend, held[1] = (held[1] if held[1] is not None else resend), None
                        send_round(i, k)
                        continue
                    resend = None
                    with cond:
                        results[k] = msg
                        if msg[0] != 'done':
                            state['stop'] = True                            # (rounds behind a failed one will never learn their first id)
                        cond.notify_all()
                    if msg[0] != 'done':
                        held[0] = None
                        break
                    if members:
                        held[0], held[1] = held[1], None
                    else:
                        held[0] = take_number()
                        if held[0] is not None:
                            send_round(i, held[0])
                if held[1] is not None:                                     # a round was handed out ahead of one that failed: take it back, in step with the worker
                    ahead = held[1]
                    m = answer(i, ahead)
                    if m[0] == 'counts':
                        self._conns[i].send(('drop', ahead))
                        answer(i, ahead)
                    with cond:
                        results[ahead] = ('dropped', ahead, None)
                        held[1] = None
                        cond.notify_all()
                self.spent['feeder_threads_cpu'] = self.spent.get('feeder_threads_cpu', 0.) + time.thread_time()
            except BaseException as e:              # (EOFError / OSError: the worker went away; anything else - a sequence that is not ASCII, a job
                #                                      that cannot be pickled, no memory for a round's members: the rounds this thread holds have
                #                                      numbers already and the caller waits for them - it must hear)
                gone = isinstance(e, (EOFError, OSError))
                text = ('a mapping worker went away: %r' % (e,)) if gone else traceback.format_exc()
                if not gone and any(r is not None for r in held):
                    # the worker still works on (or waits for the answer to) a round this thread will never finish: a process in that state would
                    # send its 'counts' into the NEXT call of a reused pool.  A fresh process takes its place - the pool stays in step.
                    try:
                        self._replace(i)
                    except Exception:
                        pass
                with cond:
                    for r in held:
                        if r is not None and r not in results:
                            results[r] = ('error', r, text)
                    state['stop'] = True
                    cond.notify_all()

        if os.environ.get('PEPPAN_KEEPER_PROFILE'):                           # cProfile of every thread that talks to a worker, merged, top functions appended to the file the variable names
            import cProfile
            import pstats
            plain_feeder, profiles = feeder, []

            def feeder(i):
                pr = cProfile.Profile()
                profiles.append(pr)
                pr.enable()
                try:
                    plain_feeder(i)
                finally:
                    pr.disable()
        threads = [threading.Thread(target=feeder, args=(i,), daemon=True) for i in range(max(1, min(self.n, n_rounds)))]
        for t in threads:
            t.start()
        try:
            for k in range(n_rounds):
                with cond:
                    while k not in results:
                        if not any(t.is_alive() for t in threads):          # (cannot happen - every thread reports what it holds - but a call that waits for ever is the worst outcome)
                            raise RuntimeError('MapWorkers: round %d was lost (no thread is working on it)' % k)
                        cond.wait(5.)
                    msg = results.pop(k)
                if msg[0] != 'done':
                    raise RuntimeError('MapWorkers: round %d failed in a worker:\n%s' % (k, msg[2]))
                mine = jobs[k * per_round:(k + 1) * per_round]
                if members:
                    yield mine, msg[2]
                else:
                    for job, G in zip(mine, msg[2]):
                        yield job, G
                slots.release()
        finally:
            with cond:
                state['stop'] = True
                cond.notify_all()
            for _ in threads:
                slots.release()                 # (a feeder waiting for a slot sees the flag and leaves)
            for t in threads:
                t.join()
            if os.environ.get('PEPPAN_KEEPER_PROFILE') and profiles:
                with open(os.environ['PEPPAN_KEEPER_PROFILE'], 'a') as out:      # (the variable names the file)
                    st = pstats.Stats(profiles[0], stream=out)
                    for pr in profiles[1:]:
                        st.add(pr)
                    st.sort_stats('tottime').print_stats(25)


if __name__ == '__main__':
    _serve(sys.argv[1], bytes.fromhex(os.environ.pop('PEPPAN_WORKER_KEY')))
