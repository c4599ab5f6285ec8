"""print the figures of a bench.py line that get compared from run to run:  python tests/dev/show_bench.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print("==", f)
    r = d['roofline']
    print("value %.0f  ms %.4f  roofline %.3f  of_sustained %s  path %.3f  reruns %s" % (
        d['value'], d['ms_per_step'], r['frac'], r.get('frac_of_sustained'), d['path_floor']['frac'],
        d['config'].get('searches_run_twice_in_timed_region')))
    if d.get('value_200_steps'): print("  200 steps: ms %.4f" % d['value_200_steps']['ms_per_step'])
    if d.get('box'): print("  box", {k: round(v, 2) for k, v in d['box'].items() if isinstance(v, float)})
    print("  form", d['config'].get('search_form'), d['config'].get('rows_per_head_pass'), "costs", [[a, round(b)] for a, b in d.get('head_pass_costs_us', {}).get('table', [])])
    for k in ('one_pass', 'level_loop_without_whole_tree_pass', 'pipelined', 'calibrated_tz', 'cli'):
        v = d.get(k)
        if not v: continue
        pf = v.get('path_floor_frac', (v.get('path_floor') or {}).get('frac'))
        print("  %-36s ms %.4f  floor %s  %s %s" % (k, v.get('ms_per_image', 0), None if pf is None else round(pf, 3), v.get('search_form'), v.get('rows_per_pass')))
    for p in (d.get('tz_sweep') or {}).get('points', []):
        print("  sweep q%.1f Tz %.4f ms %.4f %s U %s rows %s %s floor %.3f reruns %d" % (
            p['quantile'], p['Tz'], p['ms_per_image'], p['regions_per_level'], p['unique_per_level'], p['rows_per_pass'],
            p['search_form'], p['path_floor']['frac'], p['searches_run_twice']))
    if 'kernel_table' in d:
        print("  kernels:", ", ".join("%s@%s %.1f" % (e['kernel'], e['first_level'], e['avg_us']) for e in d['kernel_table']['rows']))
    if 'end_to_end' in d: print("  e2e ms %.3f backbone %.3f" % (d['end_to_end']['ms_per_image'], d['end_to_end']['backbone_ms']))
