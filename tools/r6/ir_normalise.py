import re,sys
txt=open(sys.argv[1]).read()
# strip metadata refs
txt=re.sub(r', !tbaa(\.struct)? !\d+','',txt)
txt=re.sub(r', !(noalias|alias\.scope|llvm\.loop|range|noundef|invariant\.load|prof|amdgpu\.[a-z.]+|llvm\.access\.group) !\d+','',txt)
txt=re.sub(r', !noundef !\d+','',txt)
m={}
def r(mo):
    k=mo.group(0)
    if k not in m: m[k]='%v'+str(len(m))
    return m[k]
# labels "123:" -> rename too
lines=[]
for l in txt.split('\n'):
    mo=re.match(r'^(\d+):(.*)$',l)
    if mo:
        k='%'+mo.group(1)
        if k not in m: m[k]='%v'+str(len(m))
        l=m[k][1:]+':'
    else:
        l=re.sub(r'%\d+',r,l)
    lines.append(l)
open(sys.argv[2],'w').write('\n'.join(lines))
