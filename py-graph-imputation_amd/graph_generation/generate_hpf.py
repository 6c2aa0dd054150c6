"""
produce_hpf(conf_file): per-population `POP.freqs.gz` -> `hpf.csv` + `pop_counts_file.txt`.

Drop-in for the reference's graph_generation/generate_hpf.py:8-76 (same conf keys, same
file formats byte for byte: csv module line endings, `repr(float)` frequencies,
`pop,count_sum,ratio` lines).
"""

import gzip
import json
import pathlib

project_dir = ""  # same module-level knob as the reference (generate_hpf.py:79-80)


def produce_hpf(conf_file, quiet=False):
    with open(conf_file) as fh:
        conf = json.load(fh)
    pops = conf.get("populations")
    freq_dir = project_dir + conf.get("freq_data_dir")
    out_dir = project_dir + conf.get("graph_files_path")
    counts_path = project_dir + conf.get("pops_count_file")
    hpf_path = project_dir + conf.get("freq_file")
    pathlib.Path(out_dir).mkdir(parents=True, exist_ok=True)

    if not quiet:
        bar = "*" * 100
        print(bar)
        print("Conversion to HPF file based on following configuration:")
        print("\tPopulation: {}".format(pops))
        print("\tFrequency File Directory: {}".format(freq_dir))
        print("\tOutput File: {}".format(hpf_path))
        print(bar)

    table = {}  # "POP-haplotype" -> freq, insertion ordered; a repeated key keeps its first slot
    totals = []
    for pop in pops:
        src = freq_dir + "/" + pop + ".freqs.gz"
        if not quiet:
            print("Reading Frequency File:\t {}".format(src))
        total = 0
        with gzip.open(src, "rb") as zf:
            for raw in zf.readlines():
                hap, count, freq = raw.decode("utf8").strip().split(",")
                if hap == "Haplo":
                    continue
                freq = float(freq)
                if freq == 0.0:
                    continue
                table[pop + "-" + hap] = freq
                total += float(count)
        totals.append(total)

    grand = sum(totals)
    with open(counts_path, "w") as fh:
        for pop, tot in zip(pops, totals):
            fh.write("{},{},{}\n".format(pop, tot, (tot / grand)))

    if not quiet:
        print("Writing hpf File:\t {}".format(hpf_path))
    with open(hpf_path, "w", newline="") as fh:
        fh.write("hap,pop,freq\r\n")
        for key, freq in table.items():
            pop, hap = key.split("-")
            fh.write("%s,%s,%r\r\n" % (hap, pop, freq))
