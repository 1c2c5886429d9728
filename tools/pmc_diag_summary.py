import re, sys
txt=open(sys.argv[1]).read()
blocks=re.split(r'\n(?=\S)', txt)
for b in blocks:
    lines=b.strip().split('\n'); name=lines[0]
    d={l.split()[0]: float(l.split()[1]) for l in lines[1:] if len(l.split())==2}
    if 'GRBM_GUI_ACTIVE' not in d or not any(k in name for k in ("k_point", "k_cam_diag", "k_schur", "k_pcg_iter")): continue
    cyc=d['GRBM_GUI_ACTIVE']/8
    print("%-40s %7.1f us  VALU busy %4.0f%%  TA busy %4.0f%%  wait %4.0f%%  waves/SIMD %.1f  valu/wave %5.0f" % (name[:40], cyc/2.34e3, 100*d.get('SQ_ACTIVE_INST_VALU',0)/(cyc*1024/4), 100*d.get('TA_TA_BUSY_sum',0)/(cyc*256), 100*d.get('SQ_WAIT_INST_ANY',0)/max(d.get('SQ_WAVE_CYCLES',1),1), d.get('SQ_WAVE_CYCLES',0)*4/(cyc*1024), d.get('SQ_INSTS_VALU',0)/max(d.get('SQ_WAVES',1),1)))
