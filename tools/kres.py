"""VGPRs / spills / occupancy of the k_seg instantiations that matter, from a hipcc -Rpass-analysis=kernel-resource-usage log.
usage: python3 tools/kres.py build.log [regex on the mangled name]"""
import re
import sys
text = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else r'k_segILi(256|2048)ELi[01]ELi(13|11|26|n1)E'
for b in re.split(r'Function Name: ', text)[1:]:
    mangled = b.split()[0]
    if not re.search(pat, mangled):
        continue
    def g(k):
        m = re.search(re.escape(k) + r': (\d+)', b)
        return m.group(1) if m else '?'
    print(f'{mangled:44s} VGPR {g("VGPRs"):>4} AGPR {g("AGPRs"):>3} spill {g("VGPRs Spill"):>3} SGPR {g("SGPRs"):>3} scratch {g("ScratchSize [bytes/lane]"):>4} '
          f'occ {g("Occupancy [waves/SIMD]")}')
