"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: name, calls, average us, total ms."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(r['Name'][:78].ljust(78), r['Calls'].rjust(7), '%10.1f us avg' % (float(r['AverageNs']) / 1e3),
          '%8.2f ms total' % (float(r['TotalDurationNs']) / 1e6))
