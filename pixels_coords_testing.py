"""python pixels_coords_testing.py [--zip_path Z] [--output_folder O]

Preprocess entry point with the reference's flags (/root/reference/pixels_coords_testing.py:363-364) and
step order (:343-353): unzip nested case archives -> Umean + column restructure -> drop duplicate rows ->
nearest-neighbour resample of every CSV onto the height x width pixel grid.  The resample -- the hot step,
`process_csv_with_pixel_coordinates_nearest` (:151-227) -- runs on the MI355X kernel, bit-exact, for ALL
cases of the folder in one launch.  Plotting (:233-340) is out of scope.
"""
import argparse
import os
import shutil
import zipfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import pandas as pd

height, width = 64, 512      # module globals, as in the reference (:360-361)
OUT_COLUMNS = ["y_center", "z_center", "p_center", "Umean_center", "p_convert"]
IO_THREADS = min(16, os.cpu_count() or 1)


def _read_points(csv_path) -> np.ndarray:
    arr = pd.read_csv(csv_path).to_numpy(dtype=np.float64)      # columns x, y, z, p, Umean
    if arr.shape[1] != 5:
        raise ValueError(f"{csv_path}: expected 5 columns x,y,z,p,Umean, got {arr.shape[1]}")
    return arr


def _write_table(rows: np.ndarray, output_csv) -> None:
    """the bytes `pd.DataFrame(rows, columns=OUT_COLUMNS).to_csv(output_csv, index=False)` writes (reference :223-226),
    formatted by the library's host-side writer: the pandas call was ~150 ms a case, i.e. the whole preprocess time"""
    from cfd_amd import ops
    with open(output_csv, "wb") as fh:
        fh.write((",".join(OUT_COLUMNS) + "\n").encode())
        fh.write(ops.csv_format_rows(rows))


def process_csv_with_pixel_coordinates_nearest(input_csv, output_csv):
    from cfd_amd import ops
    rows = ops.resample_nearest([_read_points(input_csv)], height, width)[0]
    _write_table(rows, output_csv)
    print(f"Nearest interpolation: {output_csv}")


def process_all_csvs_with_pixel_coordinates(output_folder, cases_per_launch: int = 256):
    """every CSV of the folder, overwritten in place like the reference (:229-231), batched on the GPU"""
    from cfd_amd import ops
    files = sorted(Path(output_folder).glob("*.csv"))
    for s in range(0, len(files), cases_per_launch):
        chunk = files[s:s + cases_per_launch]
        with ThreadPoolExecutor(max_workers=IO_THREADS) as ex:       # parser and formatter both release the GIL
            out = ops.resample_nearest(list(ex.map(_read_points, chunk)), height, width)
            list(ex.map(_write_table, out, chunk))
        for f in chunk:
            print(f"Nearest interpolation: {f}")


def extract_zip(zip_path, extract_to):
    try:
        with zipfile.ZipFile(zip_path) as z:
            for member in z.namelist():
                try:
                    z.extract(member, extract_to)
                except Exception as e:   # the reference skips unreadable members (:67-71)
                    print(f"Skip error file {member}: {e}")
    except zipfile.BadZipFile:
        print(f"Error ZIP {zip_path}")


def process_zip_folder(zip_path, output_folder):
    """:77-118 -- pick postProcessing/boxUniform/<latest time>/midBox.csv of every nested case archive"""
    output_folder = Path(output_folder)
    tmp = output_folder / "unzipped"
    tmp.mkdir(parents=True, exist_ok=True)
    extract_zip(zip_path, tmp)
    for case_zip in tmp.glob("*.zip"):
        case = case_zip.stem
        extract_zip(case_zip, tmp / case)
        case_zip.unlink(missing_ok=True)
        box = tmp / case / case / "postProcessing" / "boxUniform"
        times = [d for d in box.iterdir() if d.is_dir() and d.name.replace(".", "", 1).isdigit()] if box.exists() else []
        if not times:
            print(f"boxUniform data not found for {case}")
            continue
        src = max(times, key=lambda d: float(d.name)) / "midBox.csv"
        if src.exists():
            pd.read_csv(src).to_csv(output_folder / f"{case}_midBox.csv", index=False)
    shutil.rmtree(tmp)


def calculate_umean_and_restructure(input_csv, output_csv):
    df = pd.read_csv(input_csv)
    df["Umean"] = np.sqrt(df["U_x"] ** 2 + df["U_y"] ** 2 + df["U_z"] ** 2)     # :124
    df[["x", "y", "z", "p", "Umean"]].to_csv(output_csv, index=False)


def main(zip_path, output_folder):
    output_folder = Path(output_folder)
    output_folder.mkdir(parents=True, exist_ok=True)
    process_zip_folder(zip_path, output_folder)
    processed = output_folder / "processed"
    processed.mkdir(parents=True, exist_ok=True)
    for f in output_folder.glob("*.csv"):
        calculate_umean_and_restructure(f, processed / f.name)
    for f in processed.glob("*.csv"):
        pd.read_csv(f).drop_duplicates().to_csv(f, index=False)                  # :140-145
    process_all_csvs_with_pixel_coordinates(processed)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Process and extract data from a zip file containing CSVs.")
    ap.add_argument("--zip_path", type=str, nargs="?", default="./Case_10.zip", help="Path to the zip folder.")
    ap.add_argument("--output_folder", type=str, nargs="?", default="./output_folder",
                    help="Directory where extracted and processed files will be saved.")
    a = ap.parse_args()
    main(a.zip_path, a.output_folder)
