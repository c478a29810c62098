#!/bin/bash
# smoke() as the driver runs it (its own process), then build() + smoke() in ONE process
cd /root/repo
python -c "
import __graft_entry__ as g
g.smoke()
print([l.split()[-1] for l in open('/proc/self/maps') if 'nsvd' in l and l.rstrip().endswith('.so')][:3])
"; echo "smoke alone rc=$?"
python -c "
import __graft_entry__ as g
g.build(); g.smoke()
"; echo "build+smoke rc=$?"
