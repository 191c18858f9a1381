"""N > 1 path on CPU: two gloo ranks shard 7 clips and reduce histograms / frame counts / time."""
import os
import socket

import torch
import torch.multiprocessing as mp

from flood_uav_video_segmentation_amd import shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, _, w = shard.init("gloo")
    assert (r, w) == (rank, world_size)
    clips = shard.clips_for_rank(7, r, w)
    hist = torch.zeros(3, 5, dtype=torch.int64)
    frames = 0
    for c in clips:  # stand-in for the per-clip window loop: deterministic per-clip contributions
        for (_k0, _k1) in shard.windows_of_clip(21, 5):
            hist += torch.arange(15).view(3, 5) * (c + 1)
            frames += 5
    shard.barrier()
    total, nframes, sec = shard.reduce_run(hist, frames, 1.0 + rank)
    torch.save({"clips": clips, "hist": total, "frames": nframes, "sec": sec}, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_two_ranks_shard_and_reduce(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["clips"] == [0, 2, 4, 6] and r1["clips"] == [1, 3, 5]
    assert sorted(r0["clips"] + r1["clips"]) == list(range(7))
    expect = torch.arange(15).view(3, 5) * sum(range(1, 8)) * 4  # 4 windows per 21-frame clip
    for r in (r0, r1):
        assert torch.equal(r["hist"], expect)
        assert r["frames"] == 7 * 4 * 5
        assert r["sec"] == 2.0  # MAX over ranks


def _worker_uneven(rank, world_size, port, out_dir, num_clips, num_windows):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, _, w = shard.init("gloo")
    clips = shard.clips_for_rank(num_clips, r, w)
    hist = torch.zeros(3, 5, dtype=torch.int64)
    for c in clips:
        hist += c + 1
    total, nframes, sec = shard.reduce_run(hist, 20 * len(clips), 0.25 * (rank + 1))
    # frame-window sharding of one long clip: contiguous blocks + the boundary mask every non-first block needs
    block = shard.window_block(num_windows, r, w)
    last = torch.full((4, 6), 100 + (block[-1] if len(block) else 0), dtype=torch.uint8)
    nb = shard.exchange_boundary(last, len(block) > 0)
    shard.barrier()
    torch.save({"clips": clips, "hist": total, "frames": nframes, "sec": sec, "block": list(block),
                "neighbour": None if nb is None else int(nb[0, 0]), "desc": shard.describe()}, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_four_ranks_uneven_clip_counts_and_window_blocks(tmp_path):
    """7 clips on 4 ranks (2/2/2/1 clips) and 3 windows on 4 ranks (one EMPTY block): sums, counts, MAX time, and the
    boundary mask each block pairs its first frame with -- the nearest earlier rank that had windows."""
    mp.spawn(_worker_uneven, args=(4, _free_port(), str(tmp_path), 7, 3), nprocs=4, join=True)
    rs = [torch.load(tmp_path / f"r{i}.pt") for i in range(4)]
    assert [r["clips"] for r in rs] == [[0, 4], [1, 5], [2, 6], [3]]
    for r in rs:
        assert torch.equal(r["hist"], torch.full((3, 5), sum(range(1, 8)), dtype=torch.int64))
        assert r["frames"] == 7 * 20 and r["sec"] == 1.0 and r["desc"] == ("gloo", 4)
    assert [r["block"] for r in rs] == [[], [0], [1], [2]]           # 3*r//4 .. 3*(r+1)//4
    assert [r["neighbour"] for r in rs] == [None, None, 100, 101]     # rank 1 is the first with windows: nothing before it


def test_eight_ranks_64_clips_is_the_baseline_configs4_layout():
    """BASELINE configs[4]: 64 clips on 8 GPUs = 8 clips each, every clip exactly once (pure index arithmetic)."""
    seen = []
    for r in range(8):
        mine = shard.clips_for_rank(64, r, 8)
        assert len(mine) == 8
        seen += mine
    assert sorted(seen) == list(range(64))
    blocks = [list(shard.window_block(50, r, 8)) for r in range(8)]
    assert sum(blocks, []) == list(range(50)) and max(map(len, blocks)) - min(map(len, blocks)) <= 1


def test_bench_schedule_covers_every_window_of_the_64_clips_once():
    """bench.py under N > 1 walks shard.clip_window_schedule(64, 21, 5, rank, N): over the ranks every (clip, window) of BASELINE
    configs[4] appears exactly once, a rank owns whole clips (4 consecutive windows each), and N = 1 with one clip is configs[1]."""
    for world in (1, 2, 4, 8, 3):  # 3: uneven shards (22 / 21 / 21 clips)
        seen = []
        for r in range(world):
            mine = shard.clip_window_schedule(64, 21, 5, r, world)
            assert len(mine) == 4 * len(shard.clips_for_rank(64, r, world))
            for i in range(0, len(mine), 4):
                c = mine[i][0]
                assert c % world == r and [m for m in mine[i:i + 4]] == [(c, 0, 5), (c, 5, 10), (c, 10, 15), (c, 15, 20)]
            seen += mine
        assert sorted(seen) == [(c, 5 * w, 5 * w + 5) for c in range(64) for w in range(4)]
    assert shard.clip_window_schedule(1, 21, 5, 0, 1) == [(0, 0, 5), (0, 5, 10), (0, 10, 15), (0, 15, 20)]


_FAKE_POPEN = (
    "import sys, subprocess, os\n"
    "calls = []\n"
    "class FakePopen:\n"
    "    def __init__(self, cmd, **kw):\n"
    "        calls.append((cmd, kw))\n"
    "        rc, lines, *err = SCRIPT[len(calls) - 1]\n"
    "        self.rc, self.stdout, self.stderr = rc, iter(lines), iter(err[0] if err else [])\n"
    "    def wait(self):\n"
    "        return self.rc\n"
    "subprocess.Popen = FakePopen\n"
    "os.environ.pop('WORLD_SIZE', None)\n")


def _run_launcher(script, argv, body, extra_env=None):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"SCRIPT = {script!r}\n" + _FAKE_POPEN + f"sys.argv = {argv!r}\nsys.path.insert(0, {root!r})\nimport bench\n"
            "try:\n    bench.main()\n    rc = 0\nexcept SystemExit as e:\n    rc = e.code\n" + body + "print('launcher-ok')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "launcher-ok" in r.stdout, (r.stdout, r.stderr[-2000:])
    return r


def test_bench_launches_its_own_ranks_without_touching_torch():
    """`python bench.py --gpus N` with no WORLD_SIZE (how the driver starts it): the parent only builds the
    torch.distributed.run command for N fresh child ranks, relays their stdout and returns their exit code -- it must not import
    torch (a process that has initialised the GPU must never spawn-and-replace, and the launcher has no business paying the import)."""
    body = (
        "cmd, kw = calls[0]\n"
        "assert rc == 0 and len(calls) == 1, (rc, len(calls))\n"
        "assert 'torch' not in sys.modules, 'the launcher imported torch'\n"
        "assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=8' in cmd\n"
        "assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0\n"
        "assert cmd[-6:] == ['--gpus', '8', '--steps', '7', '--warmup', '2'] and cmd[-7].endswith('bench.py')\n"
        "assert kw['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and kw['env']['FS_BENCH_LAUNCH_ATTEMPT'] == '0'\n")
    r = _run_launcher([(0, ['{"value": 1}\n'])], ['bench.py', '--gpus', '8', '--steps', '7', '--warmup', '2'], body)
    assert '{"value": 1}' in r.stdout  # rank 0's line is relayed
    # the image exports the variable: the first attempt keeps what it finds
    body1 = "assert calls[0][1]['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '1' and rc == 0\n"
    _run_launcher([(0, ['{"value": 1}\n'])], ['bench.py', '--gpus', '8'], body1, {"HSA_ENABLE_IPC_MODE_LEGACY": "1"})


def test_bench_launcher_retries_once_with_the_other_ipc_setting_only_for_an_ipc_failure_before_the_result_line():
    """The ranks died before rank 0 printed its line AND their stderr shows that RCCL could not exchange IPC handles: ONE more
    launch, fresh children, HSA_ENABLE_IPC_MODE_LEGACY flipped, the first code handed on to the second attempt's JSON line.
    Any other failure -- a Python exception, a signal / GPU fault, a failure after the line, a second failure -- is handed back
    as it is (a faulting run must not get a silent second go)."""
    ipc = ["RuntimeError: NCCL error in: ProcessGroupNCCL.cpp, unhandled cuda error\n", "hipIpcGetMemHandle: invalid argument\n"]
    body = (
        "assert rc == 0 and len(calls) == 2, (rc, len(calls))\n"
        "e0, e1 = calls[0][1]['env'], calls[1][1]['env']\n"
        "assert (e0['HSA_ENABLE_IPC_MODE_LEGACY'], e1['HSA_ENABLE_IPC_MODE_LEGACY']) == ('0', '1')\n"
        "assert (e0['FS_BENCH_LAUNCH_ATTEMPT'], e1['FS_BENCH_LAUNCH_ATTEMPT']) == ('0', '1')\n"
        "assert 'FS_BENCH_PREV_ATTEMPT_RC' not in e0 and e1['FS_BENCH_PREV_ATTEMPT_RC'] == '1' and float(e0['FS_BENCH_LAUNCHER_T0']) > 0\n"
        "assert calls[0][1]['stderr'] == subprocess.PIPE\n"
        "p0, p1 = (c[0][c[0].index('--master-port') + 1] for c in calls)\n"
        "assert calls[0][0][-2:] == calls[1][0][-2:] == ['--gpus', '8'] and 'torch' not in sys.modules\n")
    r = _run_launcher([(1, ['some rank log\n'], ipc), (0, ['{"value": 2}\n'])], ['bench.py', '--gpus', '8'], body)
    assert "starting them once more with HSA_ENABLE_IPC_MODE_LEGACY=1" in r.stderr and '{"value": 2}' in r.stdout
    assert "hipIpcGetMemHandle: invalid argument" in r.stderr  # the ranks' stderr is relayed
    assert "launch attempts returned 1 (HSA_ENABLE_IPC_MODE_LEGACY=0) then 0 (1)" in r.stderr
    # a failure AFTER the result line: no second launch, the code is handed back
    _run_launcher([(3, ['{"value": 1}\n'], ipc)], ['bench.py', '--gpus', '8'], "assert rc == 3 and len(calls) == 1, (rc, len(calls))\n")
    # two IPC failures: the second code is handed back, nothing is launched a third time
    _run_launcher([(1, [], ipc), (7, [], ipc)], ['bench.py', '--gpus', '8'], "assert rc == 7 and len(calls) == 2, (rc, len(calls))\n")
    # a failure that is NOT the IPC signature (Python exception, OOM, bad argument): the first code, no second launch
    r = _run_launcher([(1, [], ['Traceback (most recent call last):\n', 'torch.OutOfMemoryError: HIP out of memory\n']), (0, ['{"value": 9}\n'])],
                      ['bench.py', '--gpus', '8'], "assert rc == 1 and len(calls) == 1, (rc, len(calls))\n")
    assert "not relaunched" in r.stderr and '"value": 9' not in r.stdout
    # a signal / GPU fault is never retried, whatever else stderr says
    fault = ipc + ["Memory access fault by GPU node-2\n", "traceback : Signal 6 (SIGABRT) received by PID 4242\n"]
    _run_launcher([(1, [], fault), (0, ['{"value": 9}\n'])], ['bench.py', '--gpus', '8'], "assert rc == 1 and len(calls) == 1, (rc, len(calls))\n")
    _run_launcher([(-9, [], ipc), (0, ['{"value": 9}\n'])], ['bench.py', '--gpus', '8'], "assert rc == -9 and len(calls) == 1, (rc, len(calls))\n")


def test_bench_launch_check_with_eight_real_ranks_on_cpu():
    """The 8-rank launch itself, for real: `python bench.py --gpus 8 --launch-check` spawns 8 torch.distributed.run children that
    rendezvous on 127.0.0.1 over gloo, take their shard of the 64-clip schedule, and run the end-of-run collectives.  No GPU and
    no measurement: the line carries no metric."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--launch-check"], capture_output=True, text=True,
                       timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["launch_check"] is True and j["n_gpus"] == 8 and "value" not in j and "metric" not in j
    assert j["windows_per_rank"] == [32] * 8 and j["frames_of_all_shards"] == 64 * 4 * 5 and j["max_seconds"] == 8.0
    assert j["distributed"] == {"backend": "gloo", "world_size": 8} and j["launch_attempt"] == 0


def test_single_process_is_a_no_op():
    hist, frames, sec = shard.reduce_run(torch.ones(3, 5, dtype=torch.int64), 20, 0.5)
    assert frames == 20 and sec == 0.5 and int(hist.sum()) == 15
    assert shard.clips_for_rank(64, 3, 8) == list(range(3, 64, 8))
    assert shard.windows_of_clip(21, 5) == [(0, 5), (5, 10), (10, 15), (15, 20)]
    perfect = torch.tensor([[4, 4, 4, 4, 4], [4, 4, 4, 4, 4], [4, 4, 4, 4, 4]])
    assert abs(shard.miou_from_hist(perfect) - 1.0) < 1e-9
    assert shard.exchange_boundary(torch.zeros(2, 2, dtype=torch.uint8), True) is None and shard.describe() == ("none", 1)
