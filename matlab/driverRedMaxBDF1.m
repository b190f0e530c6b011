function driverRedMaxBDF1(sceneID,batch)
% driverRedMaxBDF1  The reference's entry point (matlab-diff/driverRedMaxBDF1.m) with simLoop running on an MI355X.
%
% Same signature and same scene numbers as the reference: sceneID selects the scene of scenesRedMax.m, batch = true is
% the non-interactive mode (no drawing, energies on, no plot).  scenesRedMax, redmax.Scene and the joint / body classes
% are the reference's own files; the time stepping (simLoop, newton, eval*, computeValues) is replaced by
% redmax.simLoopHip -> redmax_hip_mex -> libredmax_hip.so.  Put this directory in front of matlab-diff/ on the path.
%
%{
% Batch mode over the scenes the HIP path covers (10, 12, 13 use forces that are outside it):
for sceneID = [0:9 11 14]
	driverRedMaxBDF1(sceneID,true)
end
%}

if nargin < 1
	sceneID = 0;
end
if nargin < 2
	batch = false;
end

scene = scenesRedMax(sceneID);
scene.init();
if batch
	scene.drawHz = 0;
	scene.computeH = true;
	scene.plotH = false;
else
	scene.test();
	scene.draw();
end

fprintf('(%d) ''%s'': tEnd=%.1f, nsteps=%d, nr=%d, nm=%d\n', sceneID, scene.name, scene.tEnd, scene.nsteps, ...
	redmax.Scene.countR(), redmax.Scene.countM());

redmax.simLoopHip(scene, 1);
scene.plotEnergies(1);

end
