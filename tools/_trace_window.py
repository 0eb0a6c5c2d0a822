"""the stepping phase of a rocprofv3 kernel trace of bench.py / a TrainStep loop: everything from the first launch of a kernel that
belongs to a training step on.  What lies before it -- model construction (hundreds of small copyBuffer / fill launches: parameter
initialisation, H2D copies), the stream / hardware-queue probe -- is not part of any step."""


def step_phase_start(cur):
    """-> (t0 in the trace's clock or None when the trace holds no loss kernel, launches in front of it)"""
    ts = [r[0] for r in cur.execute("select start from kernels where name like '%pose_loss_fwd_kernel%' order by start")]
    if len(ts) < 2:
        return None, 0
    # the kernels of a steady step (between the last two loss launches), generic runtime / ATen launches aside ...
    names = [r[0] for r in cur.execute("select distinct name from kernels where start >= ? and start < ? and name not like '%rocclr%' "
                                       "and name not like 'at::native%' and name not like 'void at::native%'", (ts[-2], ts[-1]))]
    if not names:
        return None, 0
    # ... and the first time any of them was launched
    t0 = cur.execute("select min(start) from kernels where name in (%s)" % ",".join("?" * len(names)), names).fetchone()[0]
    n = cur.execute("select count(*) from kernels where start < ? and name not like '%spin_kernel%'", (t0,)).fetchone()[0]
    return t0, n
