import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    if 'rep' in d: print(d['rep'], d['ok'], [(m['batch'], m['key'], m.get('clips')) for m in d['mismatches'][:2]])
    elif 'summary' in d: print("failed %d of %d" % (d['summary']['failed'], d['summary']['reps']))
