"""rocprofv3 --kernel-trace --stats of tools/probe/prep_stages_one.py (POLEE_PREP_REPS samples) -> kernels grouped by name,
GPU milliseconds per sample: where the device time of one sample's preparation goes.  usage: prep_kernel_stats.py stats.csv reps out.csv"""
import csv, re, sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    n = re.sub(r'rocprim::ROCPRIM_\d+_NS::detail::', 'rocprim::', n)
    m = re.search(r'(radix_sort_onesweep_iteration|onesweep_histograms|lookback_scan_kernel|radix_sort_block_sort|init_lookback_scan_state|'
                  r'block_reduce_kernel|final_reduce|merge_sort_block_merge|wrapped_scan_config|wrapped_transform_config|wrapped_reduce_config|'
                  r'wrapped_radix_sort_onesweep_config|partition)', n)
    if 'rocprim' in n and m:
        return 'rocprim ' + m.group(1)
    n = re.sub(r'\(.*$', '', n).replace('void ', '').replace('polee::', '')
    return n[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    reps = int(sys.argv[2])
    agg, tot = {}, 0.0
    for r in rows:
        k, d, c = short(r['Name']), float(r['TotalDurationNs']), int(r['Calls'])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += c
        a[1] += d
        tot += d
    out = ["kernel,calls_over_%d_samples,ms_per_sample,share" % reps]
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
        out.append("%s,%d,%.3f,%.1f%%" % (k, c, d / reps / 1e6, 100 * d / tot))
    out.append("TOTAL,,%.3f,100%%" % (tot / reps / 1e6))
    open(sys.argv[3], 'w').write("\n".join(out) + "\n")
    print("\n".join(out[:24] + out[-1:]))


if __name__ == "__main__":
    main()
