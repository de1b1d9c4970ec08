echo WIN; MVX_WINDOW=1 python tests/prof_analyse.py cfg3 168 2>&1 | tail -4
echo NOWIN; python tests/prof_analyse.py cfg3 168 2>&1 | tail -4
