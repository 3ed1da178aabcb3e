import sqlite3, sys
db=sqlite3.connect(sys.argv[1])
cur=db.cursor()
sfx=[r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch','')
rows=cur.execute(f"select k.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.stream_id from rocpd_kernel_dispatch{sfx} d join rocpd_info_kernel_symbol{sfx} k on d.kernel_id=k.id order by d.start").fetchall()
idx=[i for i,r in enumerate(rows) if 'k_collect' in r[0]]
a=idx[-2]+1; b=idx[-1]+8
t0=rows[a][1]
for n,s,e,gx,gy,sid in rows[a:b]: print(f"{n[:30]:30s} units {gx//64:6d} x{gy:3d} start {(s-t0)/1e3:9.1f} dur {(e-s)/1e3:9.1f} us stream {sid}")
