"""tools/timeline.py on a synthetic rocprofv3 kernel trace: window selection by anchor, idle gaps, overlap of two queues."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEADER = '"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp"'


def row(q, name, start, end):
    return f'"KERNEL_DISPATCH","Agent 2",{q},0,1,1,1,"{name}",1,{start},{end}'


def test_timeline_reports_gaps_and_overlap(tmp_path):
    rows = [HEADER]
    t = 1000
    for step in range(4):                              # four identical "mini-batch steps" of 100 us
        rows += [row(1, "loss_kernel(Go1PpoLossArgs)", t, t + 20_000),
                 row(1, "void gemm_nt_kernel<2>(Go1PpoGemmArgs)", t + 25_000, t + 60_000),       # 5 us idle in front
                 row(3, "void gemm_nt_kernel<2>(Go1PpoGemmArgs)", t + 30_000, t + 70_000),       # overlaps the previous one by 30 us
                 row(1, "adam_kernel(float*, float*)", t + 80_000, t + 95_000)]                  # 10 us idle in front
        t += 100_000
    trace = tmp_path / "1_kernel_trace.csv"
    trace.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "timeline.py"), str(trace), "--which", "-3"],
                         capture_output=True, text=True, check=True).stdout
    lines = out.strip().splitlines()
    assert "100.0 us, 4 kernels" in lines[0]
    body = [l.split() for l in lines[2:6]]
    assert [b[0] for b in body] == ["0.0", "25.0", "30.0", "80.0"]           # starts
    assert [b[2] for b in body] == ["0.0", "5.0", "0.0", "10.0"]             # idle gap in front of each kernel
    assert body[2][3] == "3" and body[1][4].startswith("gemm_nt_kernel")     # queue id, shortened name
    # kernels 20 + 35 + 40 + 15 = 110 us; idle 5 + 10 + 5 (tail) = 20 us; overlap = 110 - (100 - 20) = 30 us
    assert "sum of kernel durations 110.0 us" in lines[-1] and "idle inside the window 20.0 us (20.0 %)" in lines[-1] and "overlap 30.0 us" in lines[-1]
