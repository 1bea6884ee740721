"""Disassemble one kernel of a catalog entry (no GPU needed): instruction histogram to stdout, listing to /tmp/<kernel>.s

    python tools/kernel_isa.py "<catalog entry>" <kernel name>
"""
import sys,os,glob,subprocess,tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qurious_amd import catalog, planning
name=sys.argv[1]; kern=sys.argv[2]
for n,src in catalog.catalog_sources():
    if n==name:
        d=tempfile.mkdtemp()
        planning.compile_to_cache(src,d)
        obj=glob.glob(d+'/*.hsaco')[0]
        out=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','-d','--no-show-raw-insn',obj],capture_output=True,text=True).stdout
        # cut kernel
        i=out.index('<'+kern+'>:')
        j=out.find('\n\n',i)
        body=out[i:j if j>0 else None]
        open('/tmp/'+kern+'.s','w').write(body)
        import collections
        c=collections.Counter(l.split()[0] for l in body.splitlines()[1:] if l.strip() and not l.strip().startswith('//') and not l.strip().startswith(';'))
        print(len(body.splitlines()), 'lines')
        for k,v in sorted(c.items(), key=lambda kv:-kv[1])[:45]: print(v,k)
