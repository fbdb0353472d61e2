"""Per-launch HBM traffic of one kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 'gemm_bf16_pring_kernel<2, 0>' out.json

Corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a
wide (16 B/lane) coalesced read stream -- the LDS-DMA operand loads of the GEMM are exactly that -- so it is doubled.
"""
import csv
import glob
import json
import os
import sys


def per_launch(directory, counter, kernel_substr):
    vals = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel_substr!r} under {directory}")
    vals.sort()
    return {"launches": len(vals), "mean_kib": sum(vals) / len(vals), "median_kib": vals[len(vals) // 2],
            "min_kib": vals[0], "max_kib": vals[-1]}


def main():
    fetch_dir, write_dir, kernel, out = sys.argv[1:5]
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else None
    fe = per_launch(fetch_dir, "FETCH_SIZE", kernel)
    wr = per_launch(write_dir, "WRITE_SIZE", kernel)
    res = {
        "kernel": kernel, "images_per_gpu": batch,
        "fetch_size": fe, "write_size": wr,
        "fetch_bytes_corrected": 2.0 * fe["mean_kib"] * 1024.0,
        "write_bytes": wr["mean_kib"] * 1024.0,
        "traffic_bytes_per_launch": 2.0 * fe["mean_kib"] * 1024.0 + wr["mean_kib"] * 1024.0,
        "correction": "FETCH_SIZE x2 (gfx950 wide-stream undercount), KiB -> bytes; WRITE_SIZE KiB -> bytes",
        "measured": {"date": __import__("datetime").date.today().isoformat(), "commit": os.environ.get("BSI_COMMIT", "unknown"),
                     "how": "two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --k 4 --steps 1 --warmup 1`"},
    }
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
