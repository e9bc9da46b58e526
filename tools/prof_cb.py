import cProfile, pstats, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gelato_amd import driver, problem
pdict, unitdict, condition, xdict = problem.make_problem("example")
objfunc, sens = driver.make_callbacks(pdict, unitdict, condition)
driver.mock_optimizer_loop(objfunc, sens, xdict, iterations=3)
pdict["gelato_amd_share_values"] = True
driver.mock_optimizer_loop(objfunc, sens, xdict, iterations=3)
pr = cProfile.Profile(); pr.enable()
driver.mock_optimizer_loop(objfunc, sens, xdict, iterations=50)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
