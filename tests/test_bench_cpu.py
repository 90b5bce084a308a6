"""bench.py's host logic that needs no GPU: cutting a rocprofv3 counter pass into phases by the marker kernels."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_counter_rows_are_cut_into_phases_by_marker_grid_size(tmp_path):
    d = tmp_path / "pmc" / "runc"
    d.mkdir(parents=True)
    cols = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
            "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value",
            "Start_Timestamp", "End_Timestamp"]
    rows, disp = [], [0]

    def add(name, grid, counters):
        disp[0] += 1
        for cn, cv in counters.items():
            rows.append({c: 0 for c in cols} | {"Dispatch_Id": disp[0], "Grid_Size": grid, "Kernel_Name": name, "Counter_Name": cn, "Counter_Value": cv})

    add("void awry::build_something(int)", 1024, {"FETCH_SIZE": 999.0})           # before any marker: no phase
    add("awry::phase_marker_kernel()", 64 * 1, {"FETCH_SIZE": 0.0})
    for _ in range(3):
        add("void awry::count_nt2_probe_kernel<false, true>(awry::DevIndex, unsigned long const*)", 524288, {"FETCH_SIZE": 100.0})
        add("void awry::count_nt2_resume_kernel<false, true>(awry::DevIndex)", 524288, {"FETCH_SIZE": 10.0})
    add("awry::phase_marker_kernel()", 64 * 60000, {"FETCH_SIZE": 0.0})          # end of phase
    add("void awry::pack_nt2_tile_kernel<false>(unsigned char const*)", 4096, {"FETCH_SIZE": 5.0})
    add("awry::phase_marker_kernel()", 64 * 2, {"FETCH_SIZE": 0.0})
    add("void awry::locate_tile_kernel<0>(awry::DevIndex)", 2048, {"FETCH_SIZE": 7.0})
    with open(d / "1_counter_collection.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols, quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        w.writerows(rows[::-1])  # order in the file does not matter: dispatch ids do
    phases = {"headline": {"id": 1, "launches": 3}, "locate": {"id": 2, "launches": 1}}
    out = bench.parse_pmc_dir(str(tmp_path / "pmc"), phases)
    assert set(out) == {"headline", "locate"}
    assert out["headline"]["counters"]["FETCH_SIZE"] == 330.0
    assert out["headline"]["kernels"]["count_nt2_probe_kernel<false, true>"]["FETCH_SIZE"] == 300.0
    assert out["locate"]["counters"]["FETCH_SIZE"] == 7.0


def test_attach_traffic_units():
    pmc = {"x": {"traffic_bytes_per_launch": 8.0e9, "source": "s", "tcc_hit_per_launch": 1.0, "tcc_miss_per_launch": 3.0}}
    e = bench.attach_traffic({}, 2.0, pmc, "x")  # 8 GB in 2 ms = 4 TB/s = half of the 8 TB/s peak
    assert abs(e["traffic_GBs"] - 4000.0) < 1e-6 and abs(e["traffic_frac_of_peak"] - 0.5) < 1e-9 and e["l2_hit_rate"] == 0.25
    assert bench.attach_traffic({}, 2.0, pmc, "missing")["traffic"] is None
