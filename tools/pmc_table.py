#!/usr/bin/env python3
"""Print mean counter values per kernel from rocprofv3 --pmc CSV directories."""
import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            short = 'k_move_tile' if 'k_move_tile' in name else 'k_move' if 'k_move' in name else 'k_advance' if 'k_advance' in name else None
            if short:
                agg[(short, r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()):
            print('%-12s %-24s n=%-3d mean=%.5g' % (k[0], k[1], len(v), sum(v) / len(v)))
