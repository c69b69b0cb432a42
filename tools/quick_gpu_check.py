import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import orb_ref_py as R
from monoorbslam3_amd import synth
from monoorbslam3_amd.extractor import ORBExtractor
ex=ORBExtractor(2000,1.2,8,20,7)
img=synth.make_frames(1,1242,375)[0]
t=time.time(); k,d=ex(img); print('gpu extract',time.time()-t,len(k))
o=R.Oracle(2000,1.2,8,20,7); ok,od,oc=o.extract(img)
print(len(ok), oc, ex.tap_level_counts(0))
print('kp eq', len(k)==len(ok) and all(np.array_equal(k[f],ok[f]) for f in ('x','y','octave','response')))
print('angle eq', len(k)==len(ok) and np.array_equal(k['angle'],ok['angle']))
print('desc eq', d.shape==od.shape and np.array_equal(d,od))
