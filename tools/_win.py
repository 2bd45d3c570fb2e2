import sys; sys.path.insert(0,'tools')
import rocpd_summary as r
r.main(sys.argv[1], 12)
print()
r.window(sys.argv[1], top=45)
